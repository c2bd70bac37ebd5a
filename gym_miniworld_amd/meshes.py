"""Wavefront OBJ / MTL meshes of the mesh entities (MeshEnt, Key, Ball: reference gym_miniworld/entity.py:100-146,
410-434) as the triangle soups the reference hands to OpenGL, and the per-mesh BVH the render kernel walks.

Follows gym_miniworld/objmesh.py:33-216 in meaning: `v / vt / vn / usemtl / f` lines only, triangles only, faces
stably sorted by material name, float32 vertex arrays [F, 3, *], the re-centring arithmetic in float32 with the
reference's own extents (`max_coords = verts.max(axis=0).min(axis=0)` for the centring - objmesh.py:164 - and the true
maximum for `max_coords` afterwards, objmesh.py:175), per-face `Kd` colour (white without a material), the default
texture `<mesh>.png` when it exists (objmesh.py:227-231).  tests/golden/meshes.json holds SHA-256 digests of the arrays the
unmodified reference builds; tests/test_meshes.py requires equality.

The six colour variants `ball_<c>.obj` / `key_<c>.obj` of the reference are byte-identical copies of `ball.obj` / `key.obj`
that differ in their `.mtl` only, so one geometry file per shape is shipped.
"""
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MESH_DIR = os.path.join(HERE, "meshes")
COLOR_NAMES = ["blue", "green", "grey", "purple", "red", "yellow"]   # sorted(COLORS), entity.py:18

_cache = {}


def _paths(mesh_name):
    """geometry file, material library, default texture of mesh `mesh_name` (objmesh.py:22, 222-236)"""
    geom = mesh_name
    for shape in ("ball", "key"):
        if mesh_name.startswith(shape + "_") and mesh_name[len(shape) + 1:] in COLOR_NAMES:
            geom = shape   # identical geometry, the colour lives in <mesh_name>.mtl
    return (os.path.join(MESH_DIR, geom + ".obj"), os.path.join(MESH_DIR, mesh_name + ".mtl"),
            os.path.join(MESH_DIR, mesh_name + ".png"))


def _tokens(line):
    return [t for t in line.rstrip(" \r\n").split(" ") if t.strip(" ") != ""]


def _load_mtl(mtl_path, tex_path):
    default = {"Kd": np.array([1, 1, 1])}
    if os.path.exists(tex_path):
        default["map_Kd"] = tex_path
    materials = {"": default}
    if not os.path.exists(mtl_path):
        return materials
    cur = None
    with open(mtl_path) as fh:
        for line in fh:
            line = line.rstrip(" \r\n")
            if line.startswith("#") or line == "":
                continue
            tk = _tokens(line)
            if tk[0] == "newmtl":
                cur = {}
                materials[tk[1]] = cur
            elif tk[0] == "Kd":
                cur["Kd"] = np.array([float(v) for v in tk[1:]])
            elif tk[0] == "map_Kd":
                cur["map_Kd"] = os.path.join(os.path.dirname(mtl_path), tk[-1])
    return materials


class Mesh:
    """verts / norms / colors [F, 3, 3], texcs [F, 3, 2] float32 in draw order; tex_path per triangle chunk"""

    def __init__(self, mesh_name):
        obj_path, mtl_path, tex_path = _paths(mesh_name)
        materials = _load_mtl(mtl_path, tex_path)
        verts, texs, normals, faces = [], [], [], []
        cur_mtl = ""
        with open(obj_path) as fh:
            for line in fh:
                line = line.rstrip(" \r\n")
                if line.startswith("#") or line == "":
                    continue
                tk = _tokens(line)
                prefix, tk = tk[0], tk[1:]
                if prefix == "v":
                    verts.append([float(v) for v in tk])
                elif prefix == "vt":
                    texs.append([float(v) for v in tk])
                elif prefix == "vn":
                    normals.append([float(v) for v in tk])
                elif prefix == "usemtl":
                    cur_mtl = tk[0] if tk[0] in materials else ""
                elif prefix == "f":
                    assert len(tk) == 3, "only triangle faces are supported"
                    faces.append(([[int(i) for i in t.split("/") if i != ""] for t in tk], cur_mtl))
        faces.sort(key=lambda f: f[1])   # stable, by material name (objmesh.py:108)
        n = len(faces)
        self.name = mesh_name
        self.verts = np.zeros((n, 3, 3), np.float32)
        self.norms = np.zeros((n, 3, 3), np.float32)
        self.texcs = np.zeros((n, 3, 2), np.float32)
        self.colors = np.zeros((n, 3, 3), np.float32)
        self.chunks = []   # (first face, end face, texture path or None)
        prev = None
        for fi, (face, mtl_name) in enumerate(faces):
            mtl = materials[mtl_name]
            if mtl_name != prev:
                if self.chunks:
                    self.chunks[-1][1] = fi
                self.chunks.append([fi, None, mtl.get("map_Kd")])
                prev = mtl_name
            color = mtl["Kd"] if mtl else np.array((1, 1, 1))
            for li, idx in enumerate(face):
                assert len(idx) in (2, 3)
                if len(idx) == 3:
                    self.verts[fi, li], self.texcs[fi, li], self.norms[fi, li] = verts[idx[0] - 1], texs[idx[1] - 1][:2], normals[idx[2] - 1]
                else:
                    self.verts[fi, li], self.norms[fi, li] = verts[idx[0] - 1], normals[idx[1] - 1]
                self.colors[fi, li] = color
        self.chunks[-1][1] = n
        v = self.verts
        min_c = v.min(axis=0).min(axis=0)
        max_c = v.max(axis=0).min(axis=0)   # sic (objmesh.py:164): the smallest of the three per-slot maxima
        mean_c = (min_c + max_c) / 2
        v[:, :, 1] -= min_c[1]
        v[:, :, 0] -= mean_c[0]
        v[:, :, 2] -= mean_c[2]
        self.min_coords = v.min(axis=0).min(axis=0)
        self.max_coords = v.max(axis=0).max(axis=0)

    @property
    def n_tris(self):
        return self.verts.shape[0]


def get(mesh_name):
    if mesh_name not in _cache:
        _cache[mesh_name] = Mesh(mesh_name)
    return _cache[mesh_name]


def mesh_ent_dims(mesh_name, height):
    """MeshEnt.__init__ (entity.py:108-128): (scale, radius) for a mesh scaled to `height`, with the very expressions - and
    hence the scalar types of the NumPy that is installed - of the reference (under NumPy >= 2 both come out float32)."""
    sx, sy, sz = get(mesh_name).max_coords
    scale = height / sy
    radius = math.sqrt(sx * sx + sz * sz) * scale
    return scale, radius


# ------------------------------------------------------------------------------------------------------------ BVH
def build_bvh(verts, leaf_size=8, octants=True):
    """Threaded (stackless) BVH over the triangles [F, 3, 3] float32 for the render kernel: nodes in depth-first order,
    node i = (lo[3], hi[3], skip, first, count): an inner node's first child is node i + 1, `skip` is where to go when the
    box is missed (or after a leaf); a leaf lists triangles perm[first : first + count].  Splits by the surface-area heuristic
    (a median split costs 2.6x the node visits on the building's long facade triangles).  Boxes are inflated by 1e-4 of
    the mesh's extent so that the float32 slab test never rejects a ray that meets one of the box's triangles.
    A threaded hierarchy fixes the order in which the children of a node are visited; with octants=True EIGHT threadings of the
    same tree are returned, one per sign pattern of the ray direction (bit a set: the direction's component a is negative): in
    threading k the child nearer along the split axis comes first, so that a hit found early culls what lies behind it.
    -> (nodes float32 [K * M, 8] with K = 8 or 1, ints stored as bit patterns; perm int32 [F]: leaf order -> triangle index in
    draw order, the same for every threading)"""
    F = verts.shape[0]
    v = verts.astype(np.float64)
    tlo, thi = v.min(axis=1), v.max(axis=1)
    cen = (tlo + thi) / 2
    pad = 1e-4 * float(max(1e-6, (v.max() - v.min())))
    tree, order = [], []   # tree[i] = [lo, hi, first, count, axis, left, right]

    def area(lo, hi):
        dd = hi - lo
        return 2.0 * (dd[..., 0] * dd[..., 1] + dd[..., 1] * dd[..., 2] + dd[..., 0] * dd[..., 2])

    def rec(idx):
        me = len(tree)
        lo, hi = tlo[idx].min(axis=0) - pad, thi[idx].max(axis=0) + pad
        tree.append([lo, hi, 0, 0, -1, -1, -1])
        n = len(idx)
        if n <= leaf_size:
            tree[me][2], tree[me][3] = len(order), n
            order.extend(int(i) for i in idx)
        else:   # surface-area heuristic over the three centroid orders (full sweep: the meshes are a few thousand triangles)
            best = None
            for ax in range(3):
                srt = idx[np.argsort(cen[idx, ax], kind="stable")]
                plo, phi = np.minimum.accumulate(tlo[srt], axis=0), np.maximum.accumulate(thi[srt], axis=0)
                slo, shi = np.minimum.accumulate(tlo[srt][::-1], axis=0)[::-1], np.maximum.accumulate(thi[srt][::-1], axis=0)[::-1]
                k = np.arange(1, n)
                cost = area(plo[:-1], phi[:-1]) * k + area(slo[1:], shi[1:]) * (n - k)
                j = int(np.argmin(cost))
                if best is None or cost[j] < best[0]:
                    best = (float(cost[j]), srt, j + 1, ax)
            _, srt, cut, ax = best
            tree[me][4] = ax
            tree[me][5] = rec(srt[:cut])    # the child at the lower coordinates of the split axis
            tree[me][6] = rec(srt[cut:])
        return me
    rec(np.arange(F))
    M = len(tree)
    K = 8 if octants else 1
    out = np.zeros((K * M, 8), np.float32)   # lo.x lo.y lo.z skip | hi.x hi.y hi.z first | count << 24   (32 B per node)
    ints = out.view(np.int32)
    for k in range(K):
        base = k * M
        pos = 0
        stack = [0]
        rows = []   # (row, tree index); skip = the row after the node's subtree, filled when the subtree is complete
        ends = {}

        def emit(t):
            nonlocal pos
            row = pos
            pos += 1
            lo, hi, first, count, ax, left, right = tree[t]
            out[base + row, 0:3] = np.nextafter(lo.astype(np.float32), np.float32(-np.inf))
            out[base + row, 4:7] = np.nextafter(hi.astype(np.float32), np.float32(np.inf))
            ints[base + row, 7] = first | (count << 24)
            if ax >= 0:
                near_first = (left, right) if not (k >> ax) & 1 else (right, left)
                for c in near_first:
                    emit(c)
            ints[base + row, 3] = pos   # next node in depth-first order after this subtree
        import sys
        lim = sys.getrecursionlimit()
        sys.setrecursionlimit(max(lim, 10000))
        try:
            emit(0)
        finally:
            sys.setrecursionlimit(lim)
    return out, np.asarray(order, np.int32)
