"""Mesh loading and compile-time mesh processing for the model compiler.

The reference never touches meshes itself: `mjcf.from_path(...)` hands the MJCF to MuJoCo's
compiler (reference call sites: so101_sim/tasks/base/so100_task.py:388-395,
so101_sim/tasks/so100_hand_over.py:161-199).  What that compiler does to a collision mesh and what
this module reproduces:

* a mesh geom collides as the convex hull of its vertices,
* mass properties of a mesh come from integrating over its closed triangle surface,
* the mesh is expressed in a frame centred at its centre of mass and aligned with its principal
  axes, and `geom_aabb` is the tight box of the hull in that frame.
"""
from __future__ import annotations

import struct

import numpy as np
from scipy.spatial import ConvexHull


def load_stl(path: str) -> tuple[np.ndarray, np.ndarray]:
    """Binary STL -> (vertices [V,3] float64 with duplicates merged, faces [F,3] int)."""
    with open(path, "rb") as f:
        raw = f.read()
    (ntri,) = struct.unpack_from("<I", raw, 80)
    if 84 + 50 * ntri != len(raw):
        raise ValueError(f"{path}: not a binary STL (size {len(raw)} vs {84 + 50 * ntri})")
    rec = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]),
                        count=ntri, offset=84)
    tri = rec["v"].astype(np.float64).reshape(-1, 3)
    verts, inv = np.unique(tri, axis=0, return_inverse=True)
    return verts, inv.reshape(-1, 3)


def load_obj(path: str) -> tuple[np.ndarray, np.ndarray]:
    """Wavefront OBJ -> (vertices [V,3], triangle faces [F,3]); polygons are fan-triangulated."""
    verts, faces = [], []
    with open(path, "r") as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                idx = [int(tok.split("/")[0]) for tok in line.split()[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                for k in range(1, len(idx) - 1):
                    faces.append((idx[0], idx[k], idx[k + 1]))
    return np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int64).reshape(-1, 3)


def load_mesh(path: str) -> tuple[np.ndarray, np.ndarray]:
    low = path.lower()
    if low.endswith(".stl"):
        return load_stl(path)
    if low.endswith(".obj"):
        return load_obj(path)
    raise ValueError(f"unsupported mesh format: {path}")


def convex_hull(verts: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Hull vertices [H,3] (in the input frame) and outward-oriented hull triangles [T,3]."""
    hull = ConvexHull(verts)
    used = hull.vertices
    remap = -np.ones(len(verts), dtype=np.int64)
    remap[used] = np.arange(len(used))
    hv = verts[used]
    tris = remap[hull.simplices]
    # orient outward using qhull's facet equations (normal . x + offset <= 0 inside)
    a, b, c = hv[tris[:, 0]], hv[tris[:, 1]], hv[tris[:, 2]]
    n = np.cross(b - a, c - a)
    flip = np.einsum("ij,ij->i", n, hull.equations[:, :3]) < 0
    tris[flip] = tris[flip][:, ::-1]
    return hv, tris


def polyhedron_mass_properties(verts: np.ndarray, faces: np.ndarray):
    """Volume, centre of mass and inertia tensor about the COM (unit density) of a closed,
    outward-oriented triangle mesh, by signed tetrahedra against the origin."""
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    det = np.einsum("ij,ij->i", a, np.cross(b, c))
    vol = det.sum() / 6.0
    com = ((a + b + c) * det[:, None]).sum(0) / (24.0 * vol)
    # second moments  int x_i x_j dV  over each tetra (0,a,b,c)
    s = a + b + c
    outer = (np.einsum("ni,nj->nij", a, a) + np.einsum("ni,nj->nij", b, b)
             + np.einsum("ni,nj->nij", c, c) + np.einsum("ni,nj->nij", s, s))
    second = (outer * det[:, None, None]).sum(0) / 120.0
    second_c = second - vol * np.outer(com, com)
    inertia = np.trace(second_c) * np.eye(3) - second_c
    return vol, com, inertia


def principal_frame(inertia: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Eigen-decomposition with descending eigenvalues and a right-handed axis set."""
    w, v = np.linalg.eigh(inertia)
    order = np.argsort(-w)
    w, v = w[order], v[:, order]
    if np.linalg.det(v) < 0:
        v[:, 2] = -v[:, 2]
    return w, v


def mat2quat(m: np.ndarray) -> np.ndarray:
    """Rotation matrix -> unit quaternion (w,x,y,z), w >= 0 branch-stable."""
    t = np.trace(m)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s])
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = np.array([(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s])
    elif m[1, 1] > m[2, 2]:
        s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = np.array([(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s])
    else:
        s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = np.array([(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s])
    return q / np.linalg.norm(q)


def quat2mat(q: np.ndarray) -> np.ndarray:
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ])


def quat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
    ])
