"""ORACLE — TEST INFRASTRUCTURE ONLY: an algorithm-independent check of narrowphase answers.

The kernel and `so101_oracle.cpp` both find the penetration of non-flat convex pairs with MPR (the same published
algorithm in fp32 / fp64), so agreement between them says nothing about how good MPR's answer is.  What mujoco >= 3.3
(GJK / EPA; reference `so101_sim/tasks/base/so100_task.py:151` adds `multiccd`) reports for a penetrating pair is the MINIMUM
TRANSLATION: the direction n and distance d such that moving geom2 by d along n just separates the geoms, with d minimal over
all directions.  That has a definition that needs no algorithm, only support functions  h_G(u) = max_{x in G} u.x :

    overlap along u      o(u) = h_1(u) + h_2(-u)            (how far geom2 must move along +u to clear geom1)
    minimum translation  d*   = min_u o(u)                   (u over the unit sphere)

A reported contact (normal n from geom1 to geom2, depth d = -dist) is therefore checked by two numbers:
    consistency   o(n) / d   >= 1, and = 1 when the plane through the reported point really supports the Minkowski difference
    minimality    d / d*_s   >= 1 up to sampling, where d*_s = min over a direction set S (the geoms' face normals, the
                              reported normal, a Fibonacci sphere and a random local search around the best) >= d*

Everything here is brute force in numpy (hull vertices from the model blob, world poses of the bodies from the caller).
Nothing under so101_sim_amd/ imports this file.
"""
from __future__ import annotations

import numpy as np

PLANE, SPHERE, CAPSULE, CYLINDER, BOX, MESH = 0, 1, 2, 3, 4, 5


def quat2mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


class Scene:
    """World-space geoms of a model (dict from so101_sim_amd.model.blob.unpack, f64) at given body poses."""

    def __init__(self, model: dict, body_pos, body_quat):
        m = model
        self.ngeom = int(m["ngeom"][0])
        self.type = np.asarray(m["geom_type"])
        self.size = np.asarray(m["geom_size"]).reshape(-1, 3)
        body = np.asarray(m["geom_body"])
        gpos = np.asarray(m["geom_pos"]).reshape(-1, 3)
        gquat = np.asarray(m["geom_quat"]).reshape(-1, 4)
        verts = np.asarray(m["mesh_vert"]).reshape(-1, 3)
        vadr, vnum = np.asarray(m["geom_vertadr"]), np.asarray(m["geom_vertnum"])
        self.p, self.R, self.v = [], [], []
        for g in range(self.ngeom):
            Rb = quat2mat(body_quat[body[g]])
            self.p.append(np.asarray(body_pos[body[g]]) + Rb @ gpos[g])
            self.R.append(Rb @ quat2mat(gquat[g]))
            self.v.append(verts[vadr[g]: vadr[g] + vnum[g]] @ self.R[g].T + self.p[g] if self.type[g] == MESH else None)

    @classmethod
    def from_oracle(cls, model: dict, oracle):
        nb = int(model["nbody"][0])
        poses = [oracle.body_pose(b) for b in range(nb)]
        return cls(model, [p for p, _ in poses], [q for _, q in poses])

    def h(self, g: int, U):
        """support values h_g(u) for directions U [k, 3] (unit)."""
        U = np.atleast_2d(U)
        t, p, R, s = self.type[g], self.p[g], self.R[g], self.size[g]
        if t == MESH:
            return (self.v[g] @ U.T).max(axis=0)
        L = U @ R                                  # directions in the geom frame
        base = U @ p
        if t == SPHERE:
            return base + s[0]
        if t == CAPSULE:
            return base + s[0] + s[1] * np.abs(L[:, 2])
        if t == CYLINDER:
            return base + s[0] * np.hypot(L[:, 0], L[:, 1]) + s[1] * np.abs(L[:, 2])
        if t == BOX:
            return base + np.abs(L) @ s
        raise ValueError("no support function for a plane")

    def overlap(self, g1: int, g2: int, U):
        """o(u) = h_1(u) + h_2(-u): translation of geom2 along +u that separates the pair (<= 0: u already separates them)."""
        U = np.atleast_2d(U)
        return self.h(g1, U) + self.h(g2, -U)

    def face_normals(self, g: int):
        t = self.type[g]
        if t == BOX:
            return np.concatenate([self.R[g].T, -self.R[g].T])
        if t in (CYLINDER, CAPSULE):
            return np.stack([self.R[g][:, 2], -self.R[g][:, 2]])
        return np.zeros((0, 3))


def fibonacci_sphere(n: int):
    k = np.arange(n) + 0.5
    phi = np.arccos(1 - 2 * k / n)
    th = np.pi * (1 + 5 ** 0.5) * k
    return np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], axis=1)


def minimum_translation(scene: Scene, g1: int, g2: int, extra=(), n_sphere: int = 2000, refine: int = 6, seed: int = 0):
    """(d*_s, u*_s): the smallest overlap found over the geoms' face normals, `extra` directions, a Fibonacci sphere, and
    `refine` rounds of 256 random perturbations around the best direction with a shrinking radius.  An upper bound of the true
    minimum translation distance (and equal to it whenever the minimiser is a face normal of one of the two geoms)."""
    cand = [scene.face_normals(g1), -scene.face_normals(g2), fibonacci_sphere(n_sphere)]
    if len(extra):
        cand.append(np.atleast_2d(extra))
    U = np.concatenate(cand)
    o = scene.overlap(g1, g2, U)
    k = int(np.argmin(o))
    best, ubest = float(o[k]), U[k]
    rng = np.random.RandomState(seed)
    radius = 0.1
    for _ in range(refine):
        P = ubest + radius * rng.normal(size=(256, 3))
        P /= np.linalg.norm(P, axis=1, keepdims=True)
        o = scene.overlap(g1, g2, P)
        k = int(np.argmin(o))
        if o[k] < best:
            best, ubest = float(o[k]), P[k]
        radius *= 0.4
    return best, ubest


def check_contacts(scene: Scene, contacts):
    """Per geom pair of `contacts` (dicts with geom1, geom2, normal, dist; the deepest contact of a pair represents it):
    dict(pair, depth, along = o(n), mtd = d*_s, consistency = o(n) / depth, minimality = depth / d*_s, plane = geom1 is a plane)."""
    deepest = {}
    for c in contacts:
        key = (c["geom1"], c["geom2"])
        if key not in deepest or c["dist"] < deepest[key]["dist"]:
            deepest[key] = c
    out = []
    for (g1, g2), c in deepest.items():
        depth = -float(c["dist"])
        n = np.asarray(c["normal"], dtype=np.float64)
        if depth <= 0:
            continue
        if scene.type[g1] == PLANE:
            nz = scene.R[g1][:, 2]
            along = float(nz @ scene.p[g1] + scene.h(g2, -nz[None])[0])          # depth of geom2's lowest point below the plane
            out.append(dict(pair=(g1, g2), depth=depth, along=along, mtd=along, consistency=along / depth, minimality=depth / along, plane=True))
            continue
        along = float(scene.overlap(g1, g2, n[None])[0])
        mtd, _ = minimum_translation(scene, g1, g2, extra=n[None])
        out.append(dict(pair=(g1, g2), depth=depth, along=along, mtd=mtd, consistency=along / depth, minimality=depth / max(mtd, 1e-12), plane=False))
    return out
