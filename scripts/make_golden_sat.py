"""Generates tests/golden/sat_cases.json by EXECUTING the reference's own OOBB code
(/root/reference/so101_sim/utils/oobb_utils.py, loaded from the reference checkout — never copied)
on seeded random box pairs.  That module's only MuJoCo dependency is four quaternion helpers; a
numpy stand-in for exactly those four functions is installed as `mujoco` so the module imports.
The vectors therefore pin the reference's SAT logic (which axes, strict vs non-strict comparisons,
transform order); the quaternion arithmetic itself is the stand-in's."""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sat_cases.json")


def _quat2mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def mju_rotVecQuat(res, vec, quat):
    res[:] = _quat2mat(np.asarray(quat, dtype=float)) @ np.asarray(vec, dtype=float)


def mju_mulQuat(res, a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    res[:] = [aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
              aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw]


def mju_negQuat(res, q):
    res[:] = [q[0], -q[1], -q[2], -q[3]]


def mju_mat2Quat(res, mat):
    m = np.asarray(mat, dtype=float).reshape(3, 3)
    t = np.trace(m)
    if t > 0:
        s = np.sqrt(t + 1) * 2
        q = [0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = np.sqrt(1 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = [(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s]
    elif m[1, 1] > m[2, 2]:
        s = np.sqrt(1 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = [(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s]
    else:
        s = np.sqrt(1 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = [(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s]
    res[:] = np.asarray(q) / np.linalg.norm(q)


def main():
    stub = types.ModuleType("mujoco")
    stub.mju_rotVecQuat, stub.mju_mulQuat, stub.mju_negQuat, stub.mju_mat2Quat = mju_rotVecQuat, mju_mulQuat, mju_negQuat, mju_mat2Quat
    stub.MjModel = stub.MjData = object
    sys.modules["mujoco"] = stub
    spec = importlib.util.spec_from_file_location("ref_oobb_utils", os.path.join(REF, "so101_sim", "utils", "oobb_utils.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    rng = np.random.RandomState(20250629)

    def rquat(max_angle=np.pi):
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        ang = rng.uniform(-max_angle, max_angle)
        return np.concatenate([[np.cos(ang / 2)], ax * np.sin(ang / 2)])

    cases = []
    for k in range(400):
        kind = k % 4
        h0, h1 = rng.uniform(0.01, 0.12, 3), rng.uniform(0.01, 0.12, 3)
        p0, q0 = rng.uniform(-0.3, 0.3, 3), rquat()
        if kind == 0:      # generic
            p1, q1 = p0 + rng.uniform(-0.25, 0.25, 3), rquat()
        elif kind == 1:    # near-axis-aligned, close to touching along one face axis
            q0 = np.array([1.0, 0, 0, 0]); q1 = rquat(0.02)
            axis = rng.randint(3)
            p1 = p0 + rng.uniform(-0.02, 0.02, 3)
            p1[axis] = p0[axis] + (h0[axis] + h1[axis]) * rng.choice([-1, 1]) * rng.uniform(0.9, 1.1)
        elif kind == 2:    # edge-edge configurations where a 15-axis SAT would separate but 6 axes do not
            q1 = rquat()
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            p1 = p0 + d * (np.linalg.norm(h0) + np.linalg.norm(h1)) * rng.uniform(0.55, 0.95)
        else:              # the task's own geometry: banana-size box vs bowl overlap box
            h0 = np.array([0.0323, 0.0439, 0.1062]); h1 = np.array([0.03, 0.03, 0.015])
            p1 = p0 + rng.uniform(-0.12, 0.12, 3); q1 = rquat(0.3)
        b0 = ref.Oobb(position=p0, rotation=q0, half_extents=h0)
        b1 = ref.Oobb(position=p1, rotation=q1, half_extents=h1)
        cases.append(dict(box0=[p0.tolist(), q0.tolist(), h0.tolist()], box1=[p1.tolist(), q1.tolist(), h1.tolist()],
                          overlap=bool(ref.overlap_oobb_oobb(b0, b1))))
    # exact-touch cases: strict inequalities => touching counts as overlap
    for axis in range(3):
        h = np.array([0.05, 0.04, 0.03]); p1 = np.zeros(3); p1[axis] = 2 * h[axis]
        ident = np.array([1.0, 0, 0, 0])
        b0 = ref.Oobb(position=np.zeros(3), rotation=ident, half_extents=h)
        b1 = ref.Oobb(position=p1, rotation=ident, half_extents=h)
        cases.append(dict(box0=[[0, 0, 0], ident.tolist(), h.tolist()], box1=[p1.tolist(), ident.tolist(), h.tolist()],
                          overlap=bool(ref.overlap_oobb_oobb(b0, b1))))
    # transform_oobb vectors
    xf = []
    for k in range(20):
        b = ref.Oobb(position=rng.uniform(-0.1, 0.1, 3), rotation=rquat(), half_extents=rng.uniform(0.01, 0.1, 3))
        t, r = rng.uniform(-0.5, 0.5, 3), rquat()
        o = ref.transform_oobb(b, t, r)
        xf.append(dict(box=[b.position.tolist(), b.rotation.tolist(), b.half_extents.tolist()], translation=t.tolist(),
                       rotation=r.tolist(), out=[o.position.tolist(), o.rotation.tolist(), o.half_extents.tolist()]))
    json.dump(dict(source="reference so101_sim/utils/oobb_utils.py executed with a numpy stand-in for 4 mujoco.mju_* helpers",
                   overlap_cases=cases, transform_cases=xf), open(OUT, "w"))
    print(len(cases), "cases,", sum(c["overlap"] for c in cases), "overlapping")


if __name__ == "__main__":
    main()
