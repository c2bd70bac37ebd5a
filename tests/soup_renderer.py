"""Brute-force software rendition of a captured reference GL call stream (tests/golden/glstream_*.json).

Independent cross-check of the oracle's renderer (float64, numpy, no portal traversal, no
knowledge of rooms): every polygon the reference handed to OpenGL is intersected with every
sample ray, back faces are culled by WINDING (as GL_CULL_FACE does), the nearest hit wins
(GL_LESS depth test), each covered polygon is shaded once per pixel at the pixel centre with
fixed-function lighting from the captured glLightfv arguments and trilinear REPEAT texturing
from the captured texcoords, and the 8 samples are averaged.  It shares only the written render
spec (sample pattern, mip chain, LOD rule) with oracle/mw_oracle.c - none of its code.
"""
import numpy as np

SAMPLE_X = np.array([1, -1, 5, -3, -5, -7, 3, 7]) / 16.0
SAMPLE_Y = np.array([-3, 3, 1, -5, 5, -1, 7, -7]) / 16.0


def _norm(v):
    return v / np.linalg.norm(v)


def _roty(deg):
    a = np.radians(deg)
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def polygons_from_stream(g):
    """-> list of dicts: verts (n,3) world, texcs (n,2) or None, normal (3,), color (3,), tex (name or None)"""
    out = []
    room_i = 0  # per room the reference draws floor, ceiling, walls (miniworld.py:390-423)
    for p in g["polys"]:
        verts = np.array(p["verts"], float)
        texcs = np.array(p["texcs"], float)
        norms = np.array(p["norms"], float)
        R, T = np.eye(3), np.zeros(3)
        for op in p["xform"]:  # glTranslatef then glRotatef (entity.py:395-397): v_world = T + R v
            if op[0] == "translate":
                T = T + R @ np.array(op[1:4])
            else:
                assert op[2:5] == [0.0, 1.0, 0.0]
                R = R @ _roty(op[1])
        verts = verts @ R.T + T
        norms = norms @ R.T
        if p["tex_on"]:   # floor polygon (N = +Y), ceiling polygon (N = -Y, absent with no_ceiling), then the wall quads
            if p["mode"] == "GL_QUADS":
                tex = g["room_tex"][room_i][0]
                room_i += 1
            else:
                tex = g["room_tex"][room_i][1 if p["norms"][0][1] > 0 else 2]
        else:
            tex = None
        n_per = {"GL_QUADS": 4, "GL_POLYGON": len(verts), "GL_TRIANGLES": 3}[p["mode"]]
        for k in range(0, len(verts), n_per):
            out.append({"verts": verts[k:k + n_per], "texcs": texcs[k:k + n_per] if tex else None,
                        "normal": norms[k], "color": np.array(p["color"], float), "tex": tex})
    return out


def _bilinear(level, s, t):
    h, w = level.shape[:2]
    uu, vv = s * w - 0.5, t * h - 0.5
    i0, j0 = np.floor(uu).astype(int), np.floor(vv).astype(int)
    a, b = (uu - i0)[:, None], (vv - j0)[:, None]
    i1, j1 = (i0 + 1) % w, (j0 + 1) % h
    i0, j0 = i0 % w, j0 % h
    L = level[..., :3].astype(float)
    return (L[j0, i0] * (1 - a) + L[j0, i1] * a) * (1 - b) + (L[j1, i0] * (1 - a) + L[j1, i1] * a) * b


def _trilinear(levels, s, t, rho2):
    nl = len(levels)
    lam = np.where(np.isfinite(rho2), 0.5 * np.log2(np.maximum(rho2, 1e-300)), nl - 1.0)
    lam = np.clip(lam, 0, nl - 1)
    l0 = np.minimum(np.floor(lam).astype(int), nl - 1)
    l1 = np.minimum(l0 + 1, nl - 1)
    fr = (lam - np.floor(lam))[:, None]
    s, t = s - np.floor(s), t - np.floor(t)
    out = np.zeros((len(s), 3))
    for l in np.unique(np.concatenate([l0, l1])):
        c = _bilinear(levels[l], s, t)
        out += np.where((l0 == l)[:, None], (1 - fr) * c, 0) + np.where((l1 == l)[:, None] & (l1 != l0)[:, None], fr * c, 0)
    return out


def visible_cubes(g, positions, W=80, H=60):
    """get_visible_ents (miniworld.py:1222-1315) on the polygon soup: the stream's ROOM polygons, then per entity position an
    axis-aligned 0.2 m cube (drawBox's six quads, same winding), depth test GL_LESS in drawing order, back faces culled.
    Returns the set of entity indices with at least one of the 8 x W x H samples passing."""
    rooms = [p for p in polygons_from_stream(g) if p["tex"] is not None]
    fovy, aspect, _, _ = g["misc"]["gluPerspective"]
    la = g["misc"]["gluLookAt"]
    eye, center, up = np.array(la[0:3]), np.array(la[3:6]), np.array(la[6:9])
    f = _norm(center - eye)
    s_ = _norm(np.cross(f, up))
    u_ = np.cross(s_, f)
    th = np.tan(np.radians(fovy) / 2)
    tw = th * aspect
    quads = [(p["verts"], -1) for p in rooms]
    for bi, pos in enumerate(positions):
        x0, x1, y0, y1, z0, z1 = pos[0] - 0.1, pos[0] + 0.1, pos[1], pos[1] + 0.2, pos[2] - 0.1, pos[2] + 0.1
        for v in ([(x1, y1, z1), (x0, y1, z1), (x0, y0, z1), (x1, y0, z1)], [(x0, y1, z0), (x1, y1, z0), (x1, y0, z0), (x0, y0, z0)],
                  [(x0, y1, z1), (x0, y1, z0), (x0, y0, z0), (x0, y0, z1)], [(x1, y1, z0), (x1, y1, z1), (x1, y0, z1), (x1, y0, z0)],
                  [(x1, y1, z1), (x1, y1, z0), (x0, y1, z0), (x0, y1, z1)], [(x1, y0, z0), (x1, y0, z1), (x0, y0, z1), (x0, y0, z0)]):
            quads.append((np.array(v, float), bi))   # opengl.py:394-444
    py, px = np.mgrid[0:H, 0:W]
    cx, cy = (px + 0.5).ravel(), (H - 1 - py + 0.5).ravel()
    seen = set()
    for k in range(8):
        wx, wy = cx + SAMPLE_X[k], cy + SAMPLE_Y[k]
        d = f[None] + s_[None] * ((2 * wx / W - 1) * tw)[:, None] + u_[None] * ((2 * wy / H - 1) * th)[:, None]
        best_t = np.full(len(cx), np.inf)
        for v, owner in quads:
            ng = _norm(np.cross(v[1] - v[0], v[2] - v[0]))
            den = d @ ng
            with np.errstate(divide="ignore", invalid="ignore"):
                t = ((v[0] - eye) @ ng) / den
                ok = (den < 0) & (t > 0) & (t < best_t)
                P = eye[None] + t[:, None] * d
                for e in range(len(v)):
                    a, b = v[e], v[(e + 1) % len(v)]
                    ok &= (np.cross(b - a, P - a) @ ng) >= -1e-9
            best_t = np.where(ok, t, best_t)
            if owner >= 0 and ok.any():
                seen.add(owner)
    return seen


def render_stream(g, textures, W=80, H=60, ortho=False):
    """textures: name -> list of mip levels (H,W,4) uint8 with row 0 = bottom.  Returns (H,W,3) uint8.
    ortho: the frame of render_top_view (glOrtho + the fixed modelview of miniworld.py:1133-1151) instead of the camera's."""
    polys = polygons_from_stream(g)
    if ortho:
        o_l, o_r, o_b, o_t, _, o_far = g["misc"]["glOrtho"]
        m = np.array(g["misc"]["glLoadMatrixf"]).reshape(4, 4).T   # column-major upload: eye = m @ world
        assert np.array_equal(m[:3, :3], [[1, 0, 0], [0, 0, -1], [0, 1, 0]]) and not m[:3, 3].any()
        minv = m[:3, :3].T
        fdir = minv @ np.array([0.0, 0.0, -1.0])   # straight down
    else:
        fovy, aspect, _, _ = g["misc"]["gluPerspective"]
        la = g["misc"]["gluLookAt"]
        eye, center, up = np.array(la[0:3]), np.array(la[3:6]), np.array(la[6:9])
        f = _norm(center - eye)
        s_ = _norm(np.cross(f, up))
        u_ = np.cross(s_, f)
        th = np.tan(np.radians(fovy) / 2)
        tw = th * aspect
    sky = np.array(g["misc"]["glClearColor"][:3])
    Lp = np.array(g["lights"]["GL_POSITION"])
    assert Lp[3] == 0.0  # directional (miniworld.py:1026)
    Ldir = _norm(Lp[:3])
    amb, dif = np.array(g["lights"]["GL_AMBIENT"][:3]), np.array(g["lights"]["GL_DIFFUSE"][:3])

    def rays(wx, wy):
        if ortho:
            return np.broadcast_to(fdir, (len(wx), 3))
        nx, ny = 2 * wx / W - 1, 2 * wy / H - 1
        return f[None] + s_[None] * (nx * tw)[:, None] + u_[None] * (ny * th)[:, None]

    def origins(wx, wy):
        if not ortho:
            return np.broadcast_to(eye, (len(wx), 3))
        e = np.stack([o_l + wx / W * (o_r - o_l), o_b + wy / H * (o_t - o_b), np.full(len(wx), o_far)], axis=1)
        return e @ minv.T

    py, px = np.mgrid[0:H, 0:W]
    cx, cy = (px + 0.5).ravel(), (H - 1 - py + 0.5).ravel()
    npx = W * H
    best_t = np.full((8, npx), np.inf)
    best_p = np.full((8, npx), -1)
    planes = []
    for pi, p in enumerate(polys):
        v = p["verts"]
        ng = _norm(np.cross(v[1] - v[0], v[2] - v[0]))  # CCW winding -> front side
        planes.append(ng)
        for k in range(8):
            d = rays(cx + SAMPLE_X[k], cy + SAMPLE_Y[k])
            o = origins(cx + SAMPLE_X[k], cy + SAMPLE_Y[k])
            den = d @ ng
            with np.errstate(divide="ignore", invalid="ignore"):
                t = ((v[0][None] - o) @ ng) / den
            ok = (den < 0) & (t > 0) & (t < best_t[k])
            P = o + t[:, None] * d
            for e in range(len(v)):
                a, b = v[e], v[(e + 1) % len(v)]
                ok &= (np.cross(b - a, P - a) @ ng) >= -1e-9
            best_t[k] = np.where(ok, t, best_t[k])
            best_p[k] = np.where(ok, pi, best_p[k])
    acc = np.zeros((npx, 3))
    dc, dx, dy = rays(cx, cy), rays(cx + 1, cy), rays(cx, cy + 1)
    oc, ox, oy = origins(cx, cy), origins(cx + 1, cy), origins(cx, cy + 1)
    acc += (best_p == -1).sum(axis=0)[:, None] * sky[None]
    for pi in np.unique(best_p[best_p >= 0]):
        p = polys[pi]
        cnt = (best_p == pi).sum(axis=0)
        idx = np.nonzero(cnt)[0]
        C = p["color"]
        lit = np.minimum(1.0, 0.2 * C + amb * C + max(float(p["normal"] @ Ldir), 0.0) * dif * C)
        if p["tex"] is None:
            col = np.broadcast_to(lit, (len(idx), 3))
        else:
            v, tc, ng = p["verts"], p["texcs"], planes[pi]
            # affine texcoord map on the polygon's plane from its first three vertices
            e1, e2 = v[1] - v[0], v[-1] - v[0]
            M = np.array([[e1 @ e1, e1 @ e2], [e1 @ e2, e2 @ e2]])

            def texc(d, o):
                with np.errstate(divide="ignore", invalid="ignore"):
                    t = ((v[0][None] - o) @ ng) / (d @ ng)
                P = o + t[:, None] * d - v[0][None]
                ab = np.linalg.solve(M, np.stack([P @ e1, P @ e2]))
                st = tc[0][None] + ab[0][:, None] * (tc[1] - tc[0])[None] + ab[1][:, None] * (tc[-1] - tc[0])[None]
                return st, t > 0
            levels = textures[p["tex"]]
            h0, w0 = levels[0].shape[:2]
            st0, ok0 = texc(dc[idx], oc[idx])
            stx, okx = texc(dx[idx], ox[idx])
            sty, oky = texc(dy[idx], oy[idx])
            r1 = ((stx[:, 0] - st0[:, 0]) * w0) ** 2 + ((stx[:, 1] - st0[:, 1]) * h0) ** 2
            r2 = ((sty[:, 0] - st0[:, 0]) * w0) ** 2 + ((sty[:, 1] - st0[:, 1]) * h0) ** 2
            rho2 = np.where(ok0 & okx & oky, np.maximum(r1, r2), np.inf)
            texel = _trilinear(levels, st0[:, 0], st0[:, 1], rho2)
            col = lit[None] * texel / 255.0
        acc[idx] += cnt[idx][:, None] * col
    img = np.clip(acc / 8.0, 0, 1)
    return np.floor(img * 255 + 0.5).astype(np.uint8).reshape(H, W, 3)


# ------------------------------------------------------------------------------------------------------------------------
# Entity tasks (tests/golden/gen_fixtures_ents.py streams): rooms + glBegin polygons (boxes, image / text frames) + mesh draws
# (pyglet vertex lists: per-vertex normals and colours -> smooth shading).  Same brute force, vectorised over blocks of polygons.
def _xform(ops):
    """glTranslatef / glScalef / glRotatef(a, 0, 1, 0) in call order -> (M 3x3, T 3): v_world = M v + T"""
    M, T = np.eye(3), np.zeros(3)
    for op in ops:
        if op[0] == "translate":
            T = T + M @ np.array(op[1:4])
        elif op[0] == "scale":
            M = M @ np.diag(op[1:4])
        else:
            assert op[2:5] == [0.0, 1.0, 0.0]
            M = M @ _roty(op[1])
    return M, T


def _lit(normals, colors, Ldir, amb, dif):
    ndl = np.maximum(normals @ Ldir, 0.0)[:, None]
    return np.minimum(1.0, 0.2 * colors + amb[None] * colors + ndl * dif[None] * colors)


def ent_stream_triangles(g, mesh_arrays):
    """-> list of dicts: verts (n, 3) world, vcol (n, 3) lit vertex colours, texcs (n, 2) or None, tex name or None.
    mesh_arrays: mesh name -> (verts, norms, texcs, colors) float32 [F, 3, *] as handed to pyglet (pinned by meshes.json).
    Vertex lighting is OpenGL's fixed function: eye-space normal = inverse-transpose(modelview) n, NOT renormalised (GL_NORMALIZE
    off): for M = R * scale that is R n / scale."""
    Lp = np.array(g["lights"]["GL_POSITION"])
    assert Lp[3] == 0.0
    Ldir = _norm(Lp[:3])
    amb, dif = np.array(g["lights"]["GL_AMBIENT"][:3]), np.array(g["lights"]["GL_DIFFUSE"][:3])
    out = []
    for p in polygons_from_stream({"polys": g["room_polys"], "room_tex": g["room_tex"]}):
        n = len(p["verts"])
        col = _lit(np.broadcast_to(p["normal"], (n, 3)), np.broadcast_to(p["color"], (n, 3)), Ldir, amb, dif)
        out.append({"verts": p["verts"], "vcol": col, "texcs": p["texcs"], "tex": p["tex"]})
    for it in g["static_items"] + g["dynamic_items"]:
        M, T = _xform(it["xform"])
        Nmat = np.linalg.inv(M).T
        if it["type"] == "poly":
            verts = np.array(it["verts"], float) @ M.T + T
            norms = np.array(it["norms"], float) @ Nmat.T
            cols = _lit(norms, np.array(it["colors"], float), Ldir, amb, dif)
            texcs = np.array(it["texcs"], float) if it["tex_on"] else None
            n_per = {"GL_QUADS": 4, "GL_TRIANGLES": 3}[it["mode"]]
            for k in range(0, len(verts), n_per):
                out.append({"verts": verts[k:k + n_per], "vcol": cols[k:k + n_per], "texcs": None if texcs is None else texcs[k:k + n_per],
                            "tex": it["tex"] if it["tex_on"] else None})
        else:
            v, nr, tc, cl = mesh_arrays[it["mesh"]]
            assert it["mode"] == "GL_TRIANGLES" and it["chunk"] == 0
            F = v.shape[0]
            verts = v.reshape(-1, 3).astype(float) @ M.T + T
            cols = _lit(nr.reshape(-1, 3).astype(float) @ Nmat.T, cl.reshape(-1, 3).astype(float), Ldir, amb, dif)
            for k in range(F):
                out.append({"verts": verts[3 * k:3 * k + 3], "vcol": cols[3 * k:3 * k + 3], "texcs": tc[k].astype(float) if it["tex_on"] else None,
                            "tex": it["tex"] if it["tex_on"] else None})
    return out


def render_ent_stream(g, textures, mesh_arrays, W=80, H=60, block=256):
    """textures: name -> mip levels (row 0 = bottom).  8 samples per pixel, nearest front-facing polygon (winding), one shade per
    (pixel, polygon) with the attributes interpolated at the PIXEL CENTRE (barycentric on the polygon's plane, extrapolated); where
    the centre ray sees the polygon's back, at the first covering sample instead."""
    polys = ent_stream_triangles(g, mesh_arrays)
    fovy, aspect, _, _ = g["misc"]["gluPerspective"]
    la = g["misc"]["gluLookAt"]
    eye, center, up = np.array(la[0:3]), np.array(la[3:6]), np.array(la[6:9])
    f = _norm(center - eye)
    s_ = _norm(np.cross(f, up))
    u_ = np.cross(s_, f)
    th = np.tan(np.radians(fovy) / 2)
    tw = th * aspect
    sky = np.array(g["misc"]["glClearColor"][:3])

    def rays(wx, wy):
        return f[None] + s_[None] * ((2 * wx / W - 1) * tw)[:, None] + u_[None] * ((2 * wy / H - 1) * th)[:, None]
    py, px = np.mgrid[0:H, 0:W]
    cx, cy = (px + 0.5).ravel(), (H - 1 - py + 0.5).ravel()
    npx = W * H
    best_t = np.full((8, npx), np.inf)
    best_p = np.full((8, npx), -1)
    # triangles (fans of the quads / polygons) in blocks: Moeller-Trumbore in float64
    tri_v0, tri_e1, tri_e2, tri_owner = [], [], [], []
    for pi, p in enumerate(polys):
        v = p["verts"]
        for k in range(1, len(v) - 1):
            tri_v0.append(v[0]); tri_e1.append(v[k] - v[0]); tri_e2.append(v[k + 1] - v[0]); tri_owner.append(pi)
    tri_v0, tri_e1, tri_e2, tri_owner = np.array(tri_v0), np.array(tri_e1), np.array(tri_e2), np.array(tri_owner)
    for k in range(8):
        d = rays(cx + SAMPLE_X[k], cy + SAMPLE_Y[k])            # [P, 3]
        for b0 in range(0, len(tri_owner), block):
            v0, e1, e2, own = tri_v0[b0:b0 + block], tri_e1[b0:b0 + block], tri_e2[b0:b0 + block], tri_owner[b0:b0 + block]
            pv = np.cross(d[None, :, :], e2[:, None, :])          # [B, P, 3]
            det = np.einsum("bk,bpk->bp", e1, pv)
            tv = eye[None] - v0                                    # [B, 3]
            with np.errstate(divide="ignore", invalid="ignore"):
                u = np.einsum("bk,bpk->bp", tv, pv) / det
                qv = np.cross(tv, e1)                              # [B, 3]
                vv = (d @ qv.T).T / det
                t = np.einsum("bk,bk->b", e2, qv)[:, None] / det
            ok = (det > 0) & (u >= -1e-12) & (vv >= -1e-12) & (u + vv <= 1 + 1e-12) & (t > 0)
            t = np.where(ok, t, np.inf)
            j = np.argmin(t, axis=0)
            tb = t[j, np.arange(npx)]
            better = tb < best_t[k]
            best_t[k] = np.where(better, tb, best_t[k])
            best_p[k] = np.where(better, own[j], best_p[k])
    acc = (best_p == -1).sum(axis=0)[:, None] * sky[None]
    dc, dx, dy = rays(cx, cy), rays(cx + 1, cy), rays(cx, cy + 1)
    first_k = np.argmax(best_p[:, None, :] == best_p[None, :, :], axis=0)   # [8, P]: first sample with the same polygon
    for pi in np.unique(best_p[best_p >= 0]):
        p = polys[pi]
        hit = best_p == pi
        cnt = hit.sum(axis=0)
        idx = np.nonzero(cnt)[0]
        v = p["verts"]
        e1, e2 = v[1] - v[0], v[-1] - v[0]
        ng = np.cross(e1, v[2] - v[0])
        kfirst = np.argmax(hit[:, idx], axis=0)
        ds = rays(cx[idx] + SAMPLE_X[kfirst], cy[idx] + SAMPLE_Y[kfirst])
        M2 = np.array([[e1 @ e1, e1 @ e2], [e1 @ e2, e2 @ e2]])

        def bary(d):
            den = d @ ng
            with np.errstate(divide="ignore", invalid="ignore"):
                t = ((v[0] - eye) @ ng) / den
            P = eye[None] + t[:, None] * d - v[0][None]
            ab = np.linalg.solve(M2, np.stack([P @ e1, P @ e2]))
            return ab, den < 0
        ab0, front0 = bary(dc[idx])
        abs_, _ = bary(ds)
        ab = np.where(front0[None], ab0, abs_)
        interp = lambda A: A[0][None] + ab[0][:, None] * (A[1] - A[0])[None] + ab[1][:, None] * (A[-1] - A[0])[None]  # noqa: E731
        col = interp(p["vcol"])
        if p["tex"] is not None:
            levels = textures[p["tex"]]
            h0, w0 = levels[0].shape[:2]
            tc = p["texcs"]
            st0 = interp(tc)
            abx, fx = bary(dx[idx])
            aby, fy = bary(dy[idx])
            stx = tc[0][None] + abx[0][:, None] * (tc[1] - tc[0])[None] + abx[1][:, None] * (tc[-1] - tc[0])[None]
            sty = tc[0][None] + aby[0][:, None] * (tc[1] - tc[0])[None] + aby[1][:, None] * (tc[-1] - tc[0])[None]
            r1 = ((stx[:, 0] - st0[:, 0]) * w0) ** 2 + ((stx[:, 1] - st0[:, 1]) * h0) ** 2
            r2 = ((sty[:, 0] - st0[:, 0]) * w0) ** 2 + ((sty[:, 1] - st0[:, 1]) * h0) ** 2
            rho2 = np.where(front0 & fx & fy, np.maximum(r1, r2), np.inf)
            col = col * _trilinear(levels, st0[:, 0], st0[:, 1], rho2) / 255.0
        acc[idx] += cnt[idx][:, None] * col
    img = np.clip(acc / 8.0, 0, 1)
    return np.floor(img * 255 + 0.5).astype(np.uint8).reshape(H, W, 3)
