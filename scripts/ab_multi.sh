#!/bin/bash
# A/B of the _variants over several workloads.  usage: scripts/ab_multi.sh [workloads...]
for wl in ${@:-putnext8192 oneroom4096 maze8192 tmaze_features8192 fourrooms16384_dr sim2real_push8192}; do echo "== $wl"; bash scripts/ab_variants.sh $wl 2>&1 | head -2; done
