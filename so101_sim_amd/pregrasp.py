"""Scripted pre-grasp state pool for the contact-heavy pick-and-place workload (BASELINE.json configs[2]).

The reference scripts such states on the CPU (examples/automated_lerobot_dataset_generator.py:180-205 drives the arm
to a pre-grasp pose above the object, closes the gripper, lifts, moves over the container and releases; task
`BananaPickAndPlace`, :413).  Here the pool is produced on the GPU through the product path itself:

  grasp half   arm at a pre-grasp pose with the open jaws straddling the banana that lies on the table; the jaw
               closes over ~20 control steps and each env is snapshotted at a random phase of the closing
               (pads and jaw tips squeezing the banana, banana pressed on the table: 15-30 contacts)
  drop half    arm parked at the home pose, banana released 6-10 cm above the bowl's target box: it falls into
               the bowl and comes to rest there, which is the reward = 1 / discount = 0 path of
               so100_hand_over.py:238-275

`GRASP_Q` was found by a numeric IK search over the arm joints (scripts/find_pregrasp_pose.py: pad midpoint at the
banana's cross-section, pad normals along world x, fingers pointing down, pad tips above the table top).
"""
from __future__ import annotations

import math

import numpy as np

GRASP_Q = np.array([1.7353, -0.9026, 1.1451, 1.1594, 2.9829, 0.9])    # jaw open 0.9 rad
HOME_Q = np.array([0.0, -1.57079, 1.57079, 1.57079, -1.57079, 0.0])   # SO100_HOME_CTRL (so100_task.py:45-47)
BANANA_REST_Z, BOWL_REST_Z = 0.42171, 0.42262                         # notebook rest heights (KAT-1)
BANANA_GRASP_XY = (0.2616, -0.008)      # body origin that puts the banana's cross-section between the pads
BOWL_CENTRE_XY = (-0.022, -0.066)      # centre of the (x1.5) bowl's hull cloud in its body frame
BOWL_BOX_XY = (-0.0255, -0.0675)        # overlap box centre in the bowl frame (so100_hand_over.py:87-93, x1.5)
CLOSE_STEPS = 20


def _yaw_quat(torch, yaw):
    z = torch.zeros_like(yaw)
    return torch.stack([torch.cos(0.5 * yaw), z, z, torch.sin(0.5 * yaw)], 0)


def build_pickplace_pool(env, pool_size: int = 2048, seed: int = 0):
    """-> (qpos [20, K], qvel [18, K], ctrl [6, K]) device tensors.  Uses (and overwrites) the env's own state: call
    before the rollout, then `env.set_reset_pool(*pool)` and `env.reset()`."""
    torch = env.torch
    dev, N = env.device, env.n_envs
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    rnd = lambda *s: torch.rand(*s, device=dev, generator=gen)
    nrm = lambda *s: torch.randn(*s, device=dev, generator=gen)
    Kg = pool_size // 2
    Kd = pool_size - Kg
    P_q = torch.zeros(20, pool_size, device=dev)
    P_v = torch.zeros(18, pool_size, device=dev)
    P_c = torch.zeros(6, pool_size, device=dev)
    grasp = torch.tensor(GRASP_Q, dtype=torch.float32, device=dev)
    home = torch.tensor(HOME_Q, dtype=torch.float32, device=dev)

    def bowl_pose(n):
        # the reference's container distribution (so100_hand_over.py:47-55) with its collision rejection: the bowl
        # (radius 0.12 around BOWL_CENTRE_XY in its frame) must clear the static puck at (-0.2, 0.1), radius 0.08
        q = torch.zeros(7, n, device=dev)
        x = torch.full((n,), -0.3, device=dev)
        y = torch.full((n,), -0.1, device=dev)
        placed = torch.zeros(n, dtype=torch.bool, device=dev)
        for _ in range(20):
            cx, cy = -0.3 + 0.1 * rnd(n), -0.1 + 0.2 * rnd(n)
            clear = torch.hypot(cx + BOWL_CENTRE_XY[0] + 0.2, cy + BOWL_CENTRE_XY[1] - 0.1) > 0.215
            take = clear & ~placed
            x, y = torch.where(take, cx, x), torch.where(take, cy, y)
            placed |= take
        q[0], q[1] = x, y
        q[2] = BOWL_REST_Z
        q[3] = 1.0
        return q

    # ---- grasp half: simulate the closing jaw on the env's own batch, chunk by chunk
    done = 0
    env.set_reset_pool(None)
    while done < Kg:
        n = min(N, Kg - done)
        q = torch.zeros(20, N, device=dev)
        q[0:6] = grasp[:, None] + 0.01 * nrm(6, N)
        q[5] = 0.9
        q[6] = BANANA_GRASP_XY[0] + 0.004 * nrm(N)
        q[7] = BANANA_GRASP_XY[1] + 0.01 * nrm(N)
        q[8] = BANANA_REST_Z
        q[9:13] = _yaw_quat(torch, 0.05 * nrm(N))
        q[13:20] = bowl_pose(N)
        env.qpos.copy_(q)
        env.qvel.zero_()
        env.warm.zero_()
        env.begin_episode()
        snap = torch.randint(3, CLOSE_STEPS + 5, (N,), device=dev, generator=gen)
        hold = q[0:6].t().contiguous()
        taken = torch.zeros(N, dtype=torch.bool, device=dev)
        for t in range(CLOSE_STEPS + 5):
            act = hold.clone()
            act[:, 5] = max(0.0, 0.9 * (1.0 - t / CLOSE_STEPS))
            env.step_tensor(act)
            sel = (snap == t) & (env.step_type == 1) & ~taken
            sel[n:] = False
            idx = torch.nonzero(sel).flatten()
            if idx.numel():
                P_q[:, done + idx] = env.qpos[:, idx]
                P_v[:, done + idx] = env.qvel[:, idx]
                P_c[:, done + idx] = env.ctrl[:, idx]
                taken |= sel
        # envs that ended early (diverged) keep a fresh copy of the start state
        miss = torch.nonzero(~taken[:n]).flatten()
        if miss.numel():
            P_q[:, done + miss] = q[:, miss]
            P_c[:, done + miss] = q[0:6, miss]
        done += n

    # ---- drop half: no simulation needed, every body is placed at rest / at the release point
    q = torch.zeros(20, Kd, device=dev)
    q[0:6] = home[:, None]
    bowl = bowl_pose(Kd)
    q[13:20] = bowl
    q[6] = bowl[0] + BOWL_BOX_XY[0] + 0.01 * nrm(Kd)
    q[7] = bowl[1] + BOWL_BOX_XY[1] + 0.01 * nrm(Kd)
    q[8] = BOWL_REST_Z + 0.06 + 0.04 * rnd(Kd)
    q[9:13] = _yaw_quat(torch, math.pi * (2 * rnd(Kd) - 1))
    P_q[:, Kg:] = q
    P_c[:, Kg:] = home[:, None]
    return P_q, P_v, P_c
