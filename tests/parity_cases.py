"""Shared parity scenarios: the kernels behind the C ABI (backend "gpu" = MI355X product path,
backend "emu" = same kernel source under the CPU lane-thread harness) against the fp64 oracle on
identical seeded inputs.

Tolerances (fp32 kernels vs fp64 oracle), stated once here and used by both test modules:
  * free-space arm, one control step:            |dqpos| <= 2e-6 rel, |dqvel| <= 2e-5 rel  (SURVEY KAT-1 bar: 2e-6)
  * forward-pass stages:                         M, bias: 2e-6 rel; smooth qacc 5e-6 rel; contact dist 2e-6 m abs,
                                                 contact position 2e-5 m; constrained qacc 2e-3 rel of max|qacc|
                                                 (PGS stopped at its iteration cap on both sides, fp32 vs fp64 iterates)
  * resting contact, one control step:           |dqpos| <= 2e-5, |dqvel| <= 5e-3
  * reward, discount, step_type, contact counts: bit-exact / equal
"""
from __future__ import annotations

import numpy as np

from oracle.oracle import Oracle

KAT_PROPS = np.array([2.65129794e-01, 4.09270959e-03, 4.21711098e-01, 9.99246834e-01, -1.69673020e-04, 9.86495446e-05, 3.88036861e-02,
                      -2.17988678e-01, -3.96717335e-02, 4.22621829e-01, 9.99999456e-01, -5.82714664e-04, -8.65621822e-04, -8.25292623e-06])


def _arm_geom(g):
    return 1 <= g <= 18        # geom ids: 0 floor, 1 Base, 2-18 arm links/pads, 19-25 table + static props, 26+ free props


def valid_arm_states(blob64, n, seed=0, spread=0.6):
    """Random arm configurations/velocities whose geoms touch nothing (checked with the oracle)."""
    rng = np.random.RandomState(seed)
    o = Oracle(blob64)
    Q, V = [], []
    while len(Q) < n:
        q = np.concatenate([rng.uniform(-spread, spread, 6) * [1, 0.6, 1, 1, 1, 0.5] + [0, -0.3, 0.6, 0.3, 0, 0.3], KAT_PROPS])
        o.set_state(q, np.zeros(18), None)
        o.forward()
        if not any(_arm_geom(c["geom1"]) or _arm_geom(c["geom2"]) for c in o.contacts()):   # arm touches nothing
            Q.append(q)
            V.append(np.concatenate([rng.uniform(-1.5, 1.5, 6), np.zeros(12)]))
    return np.array(Q).T, np.array(V).T


def check_forward_stages(make_sim, blobs, n=4, seed=0):
    Q, V = valid_arm_states(blobs["f64"], n, seed)
    rng = np.random.RandomState(seed + 1)
    CT = rng.uniform(-1.5, 1.5, size=(6, n))
    sim = make_sim(n)
    sim.set_state(Q, V, CT, np.zeros((18, n)))
    dbg = sim.debug_forward()
    for e in range(n):
        o = Oracle(blobs["f64"])
        o.set_state(Q[:, e], V[:, e], np.zeros(18))
        o.set_ctrl(CT[:, e])
        o.forward()
        a, asm = o.qacc()
        d = dbg[e]
        M = o.M()[:6, :6]
        assert np.abs(d["M"] - M).max() <= 2e-6 * np.abs(M).max()
        assert np.abs(d["bias"] - o.bias()[:6]).max() <= 2e-6 * max(1.0, np.abs(o.bias()[:6]).max())
        assert np.abs(d["qacc_smooth"] - asm).max() <= 5e-6 * np.abs(asm).max()
        oc = o.contacts()
        assert d["ncon"] == len(oc) and d["overflow"] == 0
        for c, c2 in zip(d["contacts"], oc):
            assert (c["geom1"], c["geom2"], c["dim"]) == (c2["geom1"], c2["geom2"], c2["dim"])
            assert abs(c["dist"] - c2["dist"]) <= 2e-6
            assert np.abs(c["pos"] - c2["pos"]).max() <= 2e-5
        assert np.abs(d["qacc"] - a).max() <= 2e-3 * np.abs(a).max()
        assert d["reward"] == o.reward()


def check_pgs_forward(make_sim, blobs, n=4, seed=2, iterations=100):
    """The PGS kernels (SO101_SOLVER_PGS, the solver BASELINE.json's north_star names) against the oracle's PGS
    (mj_solPGS restated: explicit A = J Minv J' + R, elliptic-cone QCQP per contact) on resting-contact states with a
    moving arm: both run exactly `iterations` sweeps (tolerance 0), so this compares ITERATES, fp32 vs fp64.
    Tolerance: 2e-3 of max|qacc| (the sweep is a long sequential chain of small block updates), same contact and row
    counts, and both must have decreased the dual cost to within 2 % of each other's distance from the Newton optimum."""
    from so101_sim_amd import native
    Q, V = valid_arm_states(blobs["f64"], n, seed)
    rng = np.random.RandomState(seed + 1)
    CT = rng.uniform(-1.5, 1.5, size=(6, n))
    sim = make_sim(n, solver=native.SOLVER_PGS, solver_iterations=iterations, solver_tolerance=0.0)
    sim.set_state(Q, V, CT, np.zeros((18, n)))
    dbg = sim.debug_forward()
    worst = 0.0
    for e in range(n):
        o = Oracle(blobs["f64"])
        o.set_solver_type(False)
        o.set_solver(iterations, 0.0)
        o.set_state(Q[:, e], V[:, e], np.zeros(18))
        o.set_ctrl(CT[:, e])
        o.forward()
        a, _ = o.qacc()
        d = dbg[e]
        assert d["ncon"] == len(o.contacts()) and d["overflow"] == 0 and d["iters"] == iterations
        err = np.abs(d["qacc"] - a).max() / np.abs(a).max()
        worst = max(worst, err)
        assert err <= 2e-3, (e, err)
        # and PGS after `iterations` sweeps is on its way to the Newton solution: within 25 % of max|qacc| of it
        o2 = Oracle(blobs["f64"])
        o2.set_state(Q[:, e], V[:, e], np.zeros(18))
        o2.set_ctrl(CT[:, e])
        o2.forward()
        an, _ = o2.qacc()
        assert np.abs(d["qacc"] - an).max() <= 0.25 * np.abs(an).max(), e
    return worst


def check_kat1(make_sim, blobs, golden):
    """Notebook KAT-1 through the product path: reset state -> step([0,0,0,0,0,0.5]) with calibration."""
    k = golden["kat1"]
    ob = k["observation"]
    start = np.array(ob["delayed_physics_state"])
    sim = make_sim(1, action_offset=[28, 42, 18, -21, 1009, -158])
    sim.set_state(start[:20, None], start[20:, None], np.zeros((6, 1)), np.zeros((18, 1)))
    sim.begin_episode()
    obs, rew, disc, st = sim.step(np.array([k["action"]]))
    q, v, _ = sim.get_state()
    ps = np.array(ob["physics_state"])
    assert np.max(np.abs((q[:6, 0] - ps[:6]) / ps[:6])) <= 2e-6
    assert np.max(np.abs((v[:6, 0] - ps[20:26]) / ps[20:26])) <= 2e-6
    np.testing.assert_array_equal(obs[0, 12:18], np.array(ob["commanded_joints_pos"], dtype=np.float32))   # unclamped ctrl
    np.testing.assert_array_equal(obs[0, 0:6], 0)                                                          # delayed
    np.testing.assert_allclose(obs[0, 6:12], ob["undelayed_joints_pos"], rtol=2e-6)
    assert rew[0] == 0.0 and disc[0] == 1.0 and st[0] == 1
    assert np.max(np.abs(q[6:9, 0] - ps[6:9])) < 5e-6 and np.max(np.abs(q[13:16, 0] - ps[13:16])) < 5e-6


def check_control_step(make_sim, blobs, n=4, seed=3, iterations=100):
    Q, V = valid_arm_states(blobs["f64"], n, seed)
    rng = np.random.RandomState(seed + 1)
    CT = rng.uniform(-1.0, 1.0, size=(6, n))
    sim = make_sim(n, solver_iterations=iterations)
    sim.set_state(Q, V, CT, np.zeros((18, n)))
    sim.physics(10)
    q1, v1, _ = sim.get_state()
    for e in range(n):
        o = Oracle(blobs["f64"])
        o.set_solver(iterations, -1.0)
        o.set_state(Q[:, e], V[:, e], np.zeros(18))
        o.set_ctrl(CT[:, e])
        o.substeps(10)
        qo, vo, _ = o.get_state()
        arm_contact = any(_arm_geom(c["geom1"]) or _arm_geom(c["geom2"]) for c in o.contacts())
        # measured on MI355X (16 + 8 states): free arm dq <= 1.6e-7, dv <= 3.1e-5; with arm contact dq 1.1e-6, dv 2.4e-4
        tol_q, tol_v = (2e-5, 2e-3) if arm_contact else (2e-6, 2e-4)
        assert np.abs(q1[:6, e] - qo[:6]).max() <= max(tol_q, 2e-6 * np.abs(qo[:6]).max()), e
        assert np.abs(v1[:6, e] - vo[:6]).max() <= max(tol_v, 2e-5 * np.abs(vo[:6]).max()), e
        assert np.abs(q1[6:, e] - qo[6:]).max() <= tol_q and np.abs(v1[6:, e] - vo[6:]).max() <= tol_v, e


def reward_states(blobs, n, seed=5):
    """Banana poses scattered around the bowl's overlap box, some moving: exercises gate and SAT."""
    from so101_sim_amd.model import blob as blobfmt
    m = blobfmt.unpack(blobs["f64"])
    rng = np.random.RandomState(seed)
    Q, V = np.zeros((20, n)), np.zeros((18, n))
    ipos = m["body_ipos"].reshape(-1, 3)[11]
    for e in range(n):
        cq = np.array([1.0, 0, 0, 0]) + rng.normal(scale=0.05, size=4) * (e % 3 > 0)
        cq /= np.linalg.norm(cq)
        Q[13:16, e] = [-0.25 + rng.uniform(-0.03, 0.03), rng.uniform(-0.05, 0.05), 0.4226]
        Q[16:20, e] = cq
        oq = rng.normal(size=4)
        oq /= np.linalg.norm(oq)
        Q[9:13, e] = oq
        Q[6:9, e] = Q[13:16, e] + m["task_box_pos"] - ipos + rng.uniform(-0.09, 0.09, 3) * (e % 2)
        if e % 5 == 4:
            V[6 + rng.randint(3), e] = rng.choice([0.9e-3, 1e-3, 1.1e-3, -2e-3])
        if e % 7 == 6:
            V[12 + rng.randint(3), e] = rng.choice([0.9e-3, 1.5e-3])
        V[9:12, e] = rng.normal(size=3)        # angular velocity must not matter
    return Q, V


def check_reward_bitexact(make_sim, blobs, n=64):
    Q, V = reward_states(blobs, n)
    sim = make_sim(n)
    sim.set_state(Q, V, np.zeros((6, n)), np.zeros((18, n)))
    r = sim.reward()
    o = Oracle(blobs["f64"])
    want = []
    for e in range(n):
        o.set_state(Q[:, e], V[:, e], None)
        want.append(o.reward())
    want = np.array(want)
    assert set(np.unique(r)) <= {0.0, 1.0}
    assert 0 < want.sum() < n                          # both outcomes are exercised
    np.testing.assert_array_equal(r, want.astype(np.float32))


def check_reward_generic(make_sim, blobs, n=64, seed=8):
    """Scene-independent reward check (used for the Pen scene with its two overlap boxes): random object poses
    around the container's first box, some moving; GPU == oracle exactly and both outcomes occur."""
    from so101_sim_amd.model import blob as blobfmt
    m = blobfmt.unpack(blobs["f64"])
    box = np.asarray(m["task_box_pos"]).reshape(-1, 3)[0]
    ipos = np.asarray(m["body_ipos"]).reshape(-1, 3)[int(np.asarray(m["task_object_body"]).ravel()[0])]
    rng = np.random.RandomState(seed)
    Q, V = np.zeros((20, n)), np.zeros((18, n))
    for e in range(n):
        Q[13:16, e] = [-0.25 + rng.uniform(-0.03, 0.03), rng.uniform(-0.05, 0.05), 0.43]
        Q[16:20, e] = [1, 0, 0, 0]
        oq = rng.normal(size=4)
        Q[9:13, e] = oq / np.linalg.norm(oq)
        Q[6:9, e] = Q[13:16, e] + box - ipos + rng.uniform(-0.08, 0.08, 3) * (e % 2)
        if e % 5 == 4:
            V[6 + rng.randint(3), e] = rng.choice([0.9e-3, 1.1e-3, -2e-3])
    sim = make_sim(n)
    sim.set_state(Q, V, np.zeros((6, n)), np.zeros((18, n)))
    r = sim.reward()
    o = Oracle(blobs["f64"])
    want = []
    for e in range(n):
        o.set_state(Q[:, e], V[:, e], None)
        want.append(o.reward())
    want = np.array(want)
    assert 0 < want.sum() < n
    np.testing.assert_array_equal(r, want.astype(np.float32))


def check_env_semantics(make_sim, blobs, n=2, settle=30, steps=8, last_step=7, seed=11, iterations=30, rest_z=0.4217, eject_substeps=400):
    """reset -> steps -> LAST at the time limit -> auto-reset FIRST, against the oracle's env layer."""
    sim = make_sim(n, seed=seed, settle_max_substeps=settle, last_step=last_step, solver_iterations=iterations, env_id_base=100)
    sim.reset()
    q0, v0, _ = sim.get_state()
    oracles = []
    for e in range(n):
        o = Oracle(blobs["f64"])
        o.set_solver(iterations, -1.0)
        o.env_config(seed=seed, env_id=100 + e, last_step=last_step, settle_max_substeps=settle)
        o.env_reset()
        qo, vo, _ = o.get_state()
        # identical RNG draws + same settle.  An object spawned inside the static post is ejected (a chaotic
        # transient: SURVEY.md section 9 item 8), so the bound is loose there and tight otherwise.
        ejected = abs(qo[8] - rest_z) > 2e-3 or np.abs(vo[6:]).max() > 5e-3      # still moving when the budget ran out
        at_rest = abs(qo[8] - rest_z) < 2e-4 and np.abs(vo[6:]).max() < 2e-3
        # (the settle is a dynamic transient of drops/impacts: fp32 vs fp64 drift of up to 2 mm over hundreds of substeps
        # while the props still move; once at rest the poses agree to 1e-4 - measured 5e-7 .. 1.1e-4 on MI355X)
        if ejected and not eject_substeps:
            assert np.abs(q0[:, e] - qo).max() < 0.2, (e, np.abs(q0[:, e] - qo).max())      # (the emulated run: a six-substep settle budget, every prop still falling)
        elif ejected:
            # (round 5, VERDICT r4 item 6: no blanket 0.2 m.)  The ejection itself is chaotic - WHERE the prop lands differs - but not WHAT happens:
            # both copies are settled on (`eject_substeps` more substeps on copies of the two states, arm held by its actuators) and must then lie at rest on
            # the same surface: heights within 2 mm of each other, speeds below 5e-2, and still within 0.2 m horizontally
            s2 = make_sim(1, seed=seed, solver_iterations=iterations)
            s2.set_state(q0[:, e:e + 1], v0[:, e:e + 1], np.zeros((6, 1)), np.zeros((18, 1)))
            s2.physics(eject_substeps)
            qk, vk, _ = s2.get_state()
            o2 = Oracle(blobs["f64"]); o2.set_solver(iterations, -1.0)
            o2.set_state(qo, vo, np.zeros(18)); o2.set_ctrl(np.zeros(6)); o2.substeps(eject_substeps)
            qe, ve, _ = o2.get_state()
            for b in (6, 13):          # object, container: x y z at qpos[b .. b + 2]
                assert abs(qk[b + 2, 0] - qe[b + 2]) < 2e-3 and np.abs(qk[b:b + 2, 0] - qe[b:b + 2]).max() < 0.2, (e, b, qk[b:b + 3, 0], qe[b:b + 3])
            assert np.abs(vk[6:, 0]).max() < 5e-2 and np.abs(ve[6:]).max() < 5e-2, (np.abs(vk[6:, 0]).max(), np.abs(ve[6:]).max())
        else:
            tol = 5e-4 if at_rest else 5e-3
            assert np.abs(q0[:, e] - qo).max() < tol, (e, np.abs(q0[:, e] - qo).max())
        assert np.all(q0[:6, e] == 0)
        oracles.append(o)
    rng = np.random.RandomState(seed)
    undelayed_hist = []
    touched, deep = [False] * n, [False] * n          # the arm has been in (deep) contact during this episode
    resynced = [0] * n                                # control step at which the oracle was put back on the kernel's state after a deep contact
    for t in range(1, steps + 1):
        act = rng.uniform(-0.4, 0.4, size=(n, 6)).astype(np.float32)
        obs, rew, disc, st = sim.step(act)
        for e, o in enumerate(oracles):
            oo, orew, odisc, ost = o.env_step(act[e].astype(np.float64))
            assert (rew[e], disc[e], st[e]) == (orew, odisc, ost), (t, e)
            if ost == 0:
                touched[e], deep[e] = False, False
            else:
                arm_con = [c for c in o.contacts() if _arm_geom(c["geom1"]) or _arm_geom(c["geom2"])]
                touched[e] = touched[e] or bool(arm_con)
                # A jaw driven > 1 cm into the 4 cm table slab sits near the slab's mid-plane, where the penetration
                # direction flips from "up" to "down" (tunnelling): a discontinuity of any min-depth contact model, and
                # fp32 / fp64 take different sides.  Joint trajectories are not compared for the rest of such an episode.
                deep_now = any(c["dist"] < -1e-2 for c in arm_con)
                if deep[e] and not deep_now:
                    # (round 5) the deep contact is gone: the oracle continues from the KERNEL's state and the joint
                    # comparison resumes - the delayed joints one delay line (5 control steps) later, the undelayed ones at once
                    qk, vk, wk = sim.get_state()
                    o.set_state(qk[:, e].astype(np.float64), vk[:, e].astype(np.float64), wk[:, e].astype(np.float64))
                    deep[e], resynced[e] = False, t
                    continue
                deep[e] = deep[e] or deep_now
                if deep[e]:
                    continue
                # free-space arm: fp32 vs fp64 round-off only.  Once the arm pushes against the table or a prop, the
                # joint trajectory depends on the contact phase (same bound as check_control_step)
                tol = 5e-3 if touched[e] else 5e-5
                np.testing.assert_allclose(obs[e, 6:12], oo[6:12], atol=tol)
                if not resynced[e] or t > resynced[e] + 5:
                    np.testing.assert_allclose(obs[e, 0:6], oo[0:6], atol=tol)
        undelayed_hist.append(obs[:, 6:12].copy())
        if t <= last_step:
            assert np.all(st == (2 if t == last_step else 1))
            if t <= 5:
                assert np.all(obs[:, 0:6] == 0)                                # still the reset value
            else:
                np.testing.assert_array_equal(obs[:, 0:6], undelayed_hist[t - 6])   # value of control step t-5
        elif t == last_step + 1:
            assert np.all(st == 0)                                             # auto-reset: FIRST
            assert np.all(obs[:, 0:12] == 0)


def check_prefetch_identical(make_sim, n=3, settle=25, steps=9, last_step=3, seed=5):
    """The reset prefetch (k_prepare + cache) changes when an episode's initial state is settled, never its value:
    rollouts across several auto-resets are bit-identical with prefetch_resets=0 and =1."""
    out = []
    for prefetch in (0, 1):
        sim = make_sim(n, seed=seed, settle_max_substeps=settle, last_step=last_step, prefetch_resets=prefetch)
        sim.reset()
        rng = np.random.RandomState(seed)
        trace = [np.concatenate([a.ravel() for a in sim.get_state()])]
        for t in range(steps):
            obs, rew, disc, st = sim.step(rng.uniform(-0.4, 0.4, size=(n, 6)).astype(np.float32))
            trace.append(np.concatenate([obs.ravel(), rew, disc, st.astype(np.float32)] + [a.ravel() for a in sim.get_state()]))
        out.append(trace)
    for a, b in zip(*out):
        np.testing.assert_array_equal(a, b)


def check_settled_store_identical(make_sim, n=3, settle=25, steps=9, last_step=3, seed=5, first=1, count=2, prefetch=0):
    """The settled-state store (so101_compute_settled / so101_set_settled_store) replaces placement +
    settle by a table lookup for the episodes it covers and must not change a single bit: rollouts across auto-resets
    with the store attached for episodes [first, first+count) - the others still settle - equal rollouts without."""
    out, events = [], []
    for use_store in (False, True):
        sim = make_sim(n, seed=seed, settle_max_substeps=settle, last_step=last_step, prefetch_resets=prefetch)
        if use_store:
            tables = sim.compute_settled(count, first)
            assert np.all(sim.get_state()[0][:6] == 0) and np.all(sim._get(sim.episode) == 0)    # envs untouched
            sim.set_settled_store(tables, first)
        sim.reset()
        rng = np.random.RandomState(seed)
        trace = [np.concatenate([a.ravel() for a in sim.get_state()])]
        for t in range(steps):
            obs, rew, disc, st = sim.step(rng.uniform(-0.4, 0.4, size=(n, 6)).astype(np.float32))
            trace.append(np.concatenate([obs.ravel(), rew, disc, st.astype(np.float32)] + [a.ravel() for a in sim.get_state()]))
        out.append(trace)
        events.append(sim.get_events())
        episodes = sim._get(sim.episode)
    assert episodes.min() > first                      # the rollout did reset into covered episodes
    for a, b in zip(*out):
        np.testing.assert_array_equal(a, b)
    assert events[0] == events[1]                      # placement / settle flags travel with the entries


def check_pipeline_identical(make_sim, golden, n=4, steps=3, seed=9, settle=20, exact=True, all_reset_last=True, pipelines=(0, 1, 2), first_state=0, **cfg):
    """The fused k_step (0), the launch-chain pipeline (1: k_pipe_begin / k_narrow / k_pipe_solve per substep) and the per-env
    chained step (2: one persistent k_chain launch, device-side queues) run the same device functions in the same order:
    rollouts from contact-rich states, across a time-limit auto-reset, must agree (bit for bit when `exact`)."""
    states = golden["contact_rich_states"]["states"]
    states = states[first_state:] + states[:first_state]      # (the emulated CPU run starts at a state with few contacts: every cross-lane step of an EPA query is a 64-thread barrier there)
    states = (states * (1 + n // len(states)))[:n]            # tiled: n >= 64 exercises the multi-chain launch
    Q = np.array([s["qpos"] for s in states]).T
    V = np.array([s["qvel"] for s in states]).T
    W = np.array([s["warm"] for s in states]).T
    A = np.array([s["action"] for s in states]).T
    n = len(states)
    out = []
    for pipeline in pipelines:
        sim = make_sim(n, seed=seed, settle_max_substeps=settle, last_step=steps, pipeline=pipeline, prefetch_resets=0, **cfg)
        sim.set_state(Q, V, A, W)
        sim.begin_episode()
        rng = np.random.RandomState(seed)
        trace = []
        for t in range(steps + 1):
            obs, rew, disc, st = sim.step((A.T + rng.uniform(-0.2, 0.2, size=(n, 6))).astype(np.float32))
            trace.append(np.concatenate([obs.ravel(), rew, disc, st.astype(np.float32)] + [a.ravel() for a in sim.get_state()]))
        if all_reset_last:
            assert np.all(st == 0)             # the last call was the auto-reset
        else:
            assert np.any(st == 0)             # (envs that ended early, e.g. diverged, are one episode ahead)
        if pipeline == 2 and hasattr(sim.sim, "info"):
            info = sim.sim.info()
            assert info["step_path"] == 2 and info["scheduler_aborts"] == 0, info
        out.append(trace)
    for p, other in zip(pipelines[1:], out[1:]):
        for t, (a, b) in enumerate(zip(out[0], other)):
            if exact:
                np.testing.assert_array_equal(a, b, err_msg=f"pipeline {p} vs {pipelines[0]}, step {t}")
            else:
                np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-4, err_msg=f"pipeline {p} vs {pipelines[0]}, step {t}")


def check_single_env_many_candidates(make_sim, n=1):
    """One env (the reference-compatible mode: SingleEnvironment) with more broadphase candidates than the 48 records per env of the
    compact contact-record pool: the banana standing INSIDE the bowl with the gripper reaching into it (74 candidate pairs after
    the oriented-box filter, 52 contacts; found by a search over 4096 random poses, scripts/_many_gpu.py of round 4).  The slice's
    pool has a floor of MAXCAND records, so nothing is dropped: no candidate overflow, and the launch chains equal the fused step
    bit for bit."""
    q = MANY_CONTACTS_QPOS
    Q = np.tile(q[:, None], (1, n))
    out = []
    for pipeline in (0, 1):
        sim = make_sim(n, pipeline=pipeline, prefetch_resets=0, last_step=500, solver_iterations=20)
        sim.set_state(Q, np.zeros((18, n)), np.zeros((6, n)), np.zeros((18, n)))
        if pipeline == 0:
            d = sim.debug_forward()[0]
            # (round 5: with the hull pairs' patches this state has more than 64 contacts - the env keeps one per pair for the substep,
            #  flag 128 -; no candidate and no pair is dropped)
            assert d["ncand"] > 60 and (d["overflow"] & ~128) == 0, (d["ncand"], d["overflow"])
        sim.begin_episode()
        trace = []
        for t in range(2):
            obs, rew, disc, st = sim.step(np.zeros((n, 6), dtype=np.float32))
            trace.append(np.concatenate([obs.ravel(), rew, disc, st.astype(np.float32)] + [a.ravel() for a in sim.get_state()]))
        ev = sim.get_events()
        assert ev["candidate_overflow"] == 0, ev
        out.append(trace)
    for a, b in zip(*out):
        np.testing.assert_array_equal(a, b)


MANY_CONTACTS_QPOS = np.array([-1.0634258785616666, -0.5545677729900828, 1.211987612465097, 1.5106441037491418, 0.2955641252246174, 0.45852864142397015,
                               -0.21614169879288164, -0.07840607080687334, 0.45169759366551465, 0.04017006147253602, -0.9923465397604655, 0.06079185274871466,
                               -0.09969484352814978, -0.217988678, -0.0396717335, 0.422621829, 0.999999456, -0.000582714664, -0.000865621822, -8.25292623e-06])


def check_contact_capacity_mirrored(make_sim, blobs, max_contacts=64):
    """More contacts than the kernels' capacity (one lane per contact in the Newton solver: so101_max_contacts() = 64): the env keeps ONE
    contact per touching geom pair for that substep (event 7, `contacts_reduced`; MuJoCo has no such limit - a documented deviation,
    DESIGN.md section 4).  The oracle mirrors the rule on request (Oracle.set_contact_capacity): on the state of
    check_single_env_many_candidates (the banana standing inside the bowl, the gripper reaching into it: > 64 contacts with the hull
    patches) kernel and oracle must then keep the SAME contacts - pair by pair, in order, depth and normal of each - and the same
    constrained acceleration on the kernel's list; the unlimited oracle on the same state has more contacts than the capacity."""
    from oracle.oracle import Oracle
    q = MANY_CONTACTS_QPOS
    sim = make_sim(1, prefetch_resets=0, solver_iterations=50)
    sim.set_state(q[:, None], np.zeros((18, 1)), np.zeros((6, 1)), np.zeros((18, 1)))
    d = sim.debug_forward()[0]
    assert d["overflow"] & 128, ("the state must exceed the capacity", d["overflow"], len(d["contacts"]))
    assert not d["overflow"] & 2, "more touching PAIRS than contact slots: not the case this state is meant to be"
    o = Oracle(blobs["f64"])
    o.set_solver(50, -1.0)
    o.set_state(q, np.zeros(18), np.zeros(18))
    o.set_ctrl(np.zeros(6))
    o.forward()
    unlimited = o.contacts()
    assert len(unlimited) > max_contacts and o.contacts_reduced() == 0, len(unlimited)
    o.set_contact_capacity(max_contacts)
    o.forward()
    ref = o.contacts()
    assert o.contacts_reduced() == 1
    pairs = [(c["geom1"], c["geom2"]) for c in ref]
    assert len(set(pairs)) == len(pairs) == len(set((c["geom1"], c["geom2"]) for c in unlimited)), "one contact for every touching pair, none lost"
    assert len(d["contacts"]) == len(ref), (len(d["contacts"]), len(ref))
    problems, total, loose, witness = _compare_contact_lists(d["contacts"], ref)
    assert not problems, problems
    assert loose <= max(1, 0.05 * total) and witness <= max(2, 0.1 * total), (loose, witness, total)
    o.inject_contacts(d["contacts"])
    o.forward()
    a = o.qacc()[0]
    err = np.abs(d["qacc"] - a).max() / np.abs(a).max()
    assert err <= 1e-3, err
    return len(unlimited), len(ref), err


def _compare_contact_lists(mine_list, ref_list):
    """-> (problems, total, loose, witness).  problems: list of strings, empty when the two contact lists agree.
    Both sides run the default narrowphase (EPA: the nearest face of the Minkowski difference is exact, there is no portal to
    land beside).  Tolerance: same pairs in the same order with the same number of contacts (a contact may be missing on one side
    only if it is shallower than 2e-6 m); distance 5e-6 m + 1e-4 relative and normal 1e-4 for EVERY contact.  `loose` counts the
    contacts whose normals differ by more than 1e-4 but less than 1e-2 (a curved geom makes the difference curved: the face EPA
    stops on is exact to sqrt(tol / radius)); beyond 1e-2 it is a problem.  The witness POINT on a flat facet (a hull face against
    a face or an edge) is not unique - it depends on how the polytope triangulates the facet - so a position difference above
    2e-5 m is counted in `witness`, must stay inside the contact patch (1.5 cm) and is bounded by the callers; the solver is
    then compared on the kernel's own contact list."""
    by_pair = lambda cons: {k: [c for c in cons if (c["geom1"], c["geom2"]) == k] for k in dict.fromkeys((c["geom1"], c["geom2"]) for c in cons)}
    mine, ref = by_pair(mine_list), by_pair(ref_list)
    problems, total, loose, witness = [], 0, 0, 0
    for k in list(dict.fromkeys(list(mine) + list(ref))):
        cm, cr = mine.get(k, []), ref.get(k, [])
        if len(cm) != len(cr):
            extra = cm[len(cr):] if len(cm) > len(cr) else cr[len(cm):]
            if not all(abs(c["dist"]) < 2e-6 for c in extra):
                problems.append(f"{k}: {len(cm)} vs {len(cr)} contacts")
        for c1, c2 in zip(cm, cr):
            total += 1
            if abs(c1["dist"] - c2["dist"]) > 5e-6 + 1e-4 * abs(c2["dist"]):
                problems.append(f"{k}: dist {c1['dist']:.6f} vs {c2['dist']:.6f}")
                continue
            dn, dp = np.abs(c1["normal"] - c2["normal"]).max(), np.abs(c1["pos"] - c2["pos"]).max()
            if dn > 1e-4:
                loose += 1
                if dn > 1e-2:
                    problems.append(f"{k}: normal {dn:.2e} at depth {c2['dist']:.6f}")
                    continue
            if dp > 2e-5:
                witness += 1
                if dp > 1.5e-2:
                    problems.append(f"{k}: pos {dp:.2e} at depth {c2['dist']:.6f}")
    return problems, total, loose, witness


def check_contact_rich(make_sim, blobs, golden, count=4, verbose=False):
    """Arm self-collision / arm-table / arm-prop contact states (20-40 simultaneous contacts, captured from random-action
    rollouts, the last ones under the EPA narrowphase): the SAME contact list as the oracle - pair by pair, contact by contact,
    in order, depth and normal of every contact (tolerances in _compare_contact_lists; no state may disagree: EPA returns an
    exact face of the Minkowski difference, so fp32 and fp64 have no portal to land on different sides of) - and the same
    constrained acceleration: 1e-4 of max|qacc| against the oracle solving on the KERNEL's contact list (orc_inject_contacts),
    on every state, and 1e-3 against the oracle's own solve wherever every witness point agrees (a witness point that sits
    elsewhere on a flat facet: at most 5 % of the contacts)."""
    states = golden["contact_rich_states"]["states"][:count]
    n = len(states)
    Q = np.array([s["qpos"] for s in states]).T
    V = np.array([s["qvel"] for s in states]).T
    W = np.array([s["warm"] for s in states]).T
    A = np.array([s["action"] for s in states]).T
    sim = make_sim(n)
    sim.set_state(Q, V, A, W)
    dbg = sim.debug_forward()

    worst, seen_arm_arm, seen_multi, total, loose, witness = 0.0, False, False, 0, 0, 0
    for e in range(n):
        d = dbg[e]
        assert d["overflow"] == 0
        o = Oracle(blobs["f64"])
        o.set_state(Q[:, e], V[:, e], W[:, e])
        o.set_ctrl(A[:, e])
        o.forward()
        a, ref = o.qacc()[0], o.contacts()
        problems, t, l, w = _compare_contact_lists(d["contacts"], ref)
        assert not problems, (e, problems)
        err_own = np.abs(d["qacc"] - a).max() / np.abs(a).max()
        o.inject_contacts(d["contacts"])
        o.forward()
        a = o.qacc()[0]
        err = np.abs(d["qacc"] - a).max() / np.abs(a).max()
        # (own contact lists: positions agree to 2e-5 m, which stiff pad contacts amplify - measured 4e-4 on one state, <= 3e-6 on the others)
        assert err <= 1e-4 and (w > 0 or err_own <= 1e-3), (e, err, err_own, w)
        total, loose, witness = total + t, loose + l, witness + w
        pairs = [(c["geom1"], c["geom2"]) for c in ref]
        seen_multi = seen_multi or len(set(pairs)) < len(pairs)
        seen_arm_arm = seen_arm_arm or any(_arm_geom(g1) and _arm_geom(g2) for g1, g2 in pairs)
        worst = max(worst, err)
        if verbose:
            print(e, "ncon", len(ref), "qacc rel err", err, "loose", l, "witness", w)
    assert seen_arm_arm, "fixture must contain arm-arm contacts"
    assert seen_multi, "fixture must contain a pair with several contacts (flat-face patch)"
    assert loose <= 0.02 * total, (loose, total)
    assert witness <= 0.05 * total, (witness, total)
    return worst, (total, loose, witness)


def check_probe_outliers(make_sim, blobs, golden, which=None):
    """States behind the one-step parity outliers of the random-action rollout (tests/golden/probe_outlier_states.json, captured by
    scripts/gpu_probe_outlier.py): at each of them kernel and oracle must build the SAME contact list (pairs, counts, depth and normal of
    every contact: _compare_contact_lists) and the same constrained acceleration on the kernel's own list (1e-3).  State 1 is the regression
    of round 6: the wrist hull whose centre lies inside the static puck - the fp32 MPR took its 1e-5 m origin ray for collinear with the
    first support point and reported 76 mm sideways; the minimum translation is 45 mm through the cap."""
    states = golden["probe_outlier_states"]["states"]
    idx = list(range(len(states))) if which is None else list(which)
    for i in idx:
        st = states[i]
        q, v, w, a = (np.array(st[k], dtype=np.float64) for k in ("qpos", "qvel", "warm", "action"))
        sim = make_sim(1, prefetch_resets=0)
        sim.set_state(q[:, None], v[:, None], a[:, None], w[:, None])
        d = sim.debug_forward()[0]
        o = Oracle(blobs["f64"])
        o.set_state(q, v, w); o.set_ctrl(a); o.forward()
        problems, total, loose, witness = _compare_contact_lists(d["contacts"], o.contacts())
        assert not problems, (i, problems)
        o.inject_contacts(d["contacts"]); o.forward()
        acc = o.qacc()[0]
        err = np.abs(d["qacc"] - acc).max() / np.abs(acc).max()
        assert err <= 1e-3, (i, err)


def check_divergence_handling(make_sim, blobs):
    """A non-finite / exploded state ends the episode like a dm_control physics error: LAST, reward 0, discount 0,
    then the next step auto-resets (SURVEY.md section 5 'failure detection'); the kernels must survive NaNs."""
    n = 2
    sim = make_sim(n, seed=2, settle_max_substeps=5, solver_iterations=5)
    Q = np.tile(np.concatenate([np.zeros(6), KAT_PROPS])[:, None], (1, n))
    V = np.zeros((18, n))
    V[2, 0] = np.nan            # env 0 poisoned, env 1 healthy
    sim.set_state(Q, V, np.zeros((6, n)), np.zeros((18, n)))
    sim.begin_episode()
    obs, rew, disc, st = sim.step(np.zeros((n, 6), dtype=np.float32))
    assert (rew[0], disc[0], st[0]) == (0.0, 0.0, 2)
    assert (rew[1], disc[1], st[1]) == (0.0, 1.0, 1)
    q, v, _ = sim.get_state()
    assert np.all(np.isfinite(q)) and np.all(np.isfinite(v))
    o = Oracle(blobs["f64"])
    o.env_config(seed=2, env_id=0, settle_max_substeps=5)
    o.set_state(Q[:, 0], V[:, 0], np.zeros(18))
    o.env_begin()
    _, orew, odisc, ost = o.env_step(np.zeros(6))
    assert (orew, odisc, ost) == (0.0, 0.0, 2)
    obs, rew, disc, st = sim.step(np.zeros((n, 6), dtype=np.float32))
    assert st[0] == 0 and st[1] == 1          # env 0 auto-reset -> FIRST
