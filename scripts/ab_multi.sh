for wl in putnext8192 oneroom4096 maze8192 tmaze_features8192 fourrooms16384_dr; do echo "== $wl"; bash scripts/ab_variants.sh $wl 2>&1 | head -2; done
