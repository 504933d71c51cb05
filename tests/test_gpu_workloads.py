"""MI355X tests of the BASELINE.json workloads beyond configs[1]: the contact-heavy pick-and-place pool (configs[2]),
the mixed Banana + Pen suite with per-env mass randomisation (configs[3]) and the 32768-env per-GPU share of
configs[4] - oracle comparisons on small slices, size-independent properties at the full sizes with the benchmarked
solver settings (Newton, 100 iterations, tolerance 1e-8)."""
import os

import numpy as np
import pytest

from oracle.oracle import Oracle, rng_uniform
from tests import parity_cases as pc
from tests.simharness import ArraySim

pytestmark = pytest.mark.gpu

LO = np.array([-np.pi, -3.14158, -3.14158, -3.14158, -3.14158, 0.0], dtype=np.float32)
HI = np.array([np.pi, 3.14158, 3.14158, 3.14158, 3.14158, 0.08], dtype=np.float32)


@pytest.fixture(scope="module")
def make_sim(blobs):
    def f(n, seed=0, **cfg):
        return ArraySim(blobs["f32"], n, backend="gpu", seed=seed, **cfg)
    return f


def _batched_env(name, n, **kw):
    import os
    from so101_sim_amd import task_suite
    cwd = os.getcwd()
    os.chdir("/tmp")                       # no calibration offsets
    try:
        return task_suite.create_task_env(name, time_limit=kw.pop("time_limit", 10.0), random_state=0, n_envs=n, **kw)
    finally:
        os.chdir(cwd)


# ------------------------------------------------------------------ per-env mass randomisation (configs[3])
def test_mass_scale_matches_oracle(make_sim, blobs):
    """Props' mass / inertia scaled per env: one control step from resting-contact states (arm moving freely) against the
    oracle with the same scales; an env with scale 1 is bit-identical to a run without the array."""
    n = 8
    Q, V = pc.valid_arm_states(blobs["f64"], n, seed=4)
    rng = np.random.RandomState(5)
    CT = rng.uniform(-1.0, 1.0, size=(6, n))
    scale = rng.uniform(0.5, 1.5, size=(2, n))
    scale[:, 0] = 1.0
    sim = make_sim(n, solver_iterations=100)
    sim.set_mass_scale(scale)
    sim.set_state(Q, V, CT, np.zeros((18, n)))
    sim.physics(10)
    q1, v1, _ = sim.get_state()
    ref = make_sim(1, solver_iterations=100)
    ref.set_state(Q[:, :1], V[:, :1], CT[:, :1], np.zeros((18, 1)))
    ref.physics(10)
    np.testing.assert_array_equal(ref.get_state()[0][:, 0], q1[:, 0])
    for e in range(n):
        o = Oracle(blobs["f64"])
        o.set_solver(100, -1.0)
        o.set_mass_scale(scale[:, e])
        o.set_state(Q[:, e], V[:, e], np.zeros(18))
        o.set_ctrl(CT[:, e])
        o.substeps(10)
        qo, vo, _ = o.get_state()
        assert np.abs(q1[:, e] - qo).max() <= 2e-5 and np.abs(v1[:, e] - vo).max() <= 5e-3, e
    # (a prop resting on static geometry moves the same for any mass - the soft-contact regulariser scales with
    # body_invweight0; the mass shows where the arm's bounded motor torques push it: see the pool test below)


# ------------------------------------------------------------------ reset pool (configs[2])
def test_reset_pool_semantics(make_sim):
    """so101_set_reset_pool: episode k of global env id g starts from pool entry floor(u * K), u = counter RNG at
    (seed, g, k, draw 1000); auto-resets inside step draw the next episode's entry; pool_size 0 restores the settle."""
    n, K, seed, base = 6, 7, 13, 40
    rng = np.random.RandomState(1)
    PQ = rng.uniform(-0.1, 0.1, size=(20, K)).astype(np.float32)
    PQ[6:9] += np.array([[0.25], [0.0], [0.6]], dtype=np.float32)
    PQ[13:16] += np.array([[-0.25], [0.0], [0.6]], dtype=np.float32)
    PQ[9:13] = np.array([[1.0], [0], [0], [0]], dtype=np.float32)
    PQ[16:20] = np.array([[1.0], [0], [0], [0]], dtype=np.float32)
    PV = rng.uniform(-0.01, 0.01, size=(18, K)).astype(np.float32)
    PC = rng.uniform(-0.5, 0.5, size=(6, K)).astype(np.float32)
    sim = make_sim(n, seed=seed, env_id_base=base, last_step=2, settle_max_substeps=10)
    sim.set_reset_pool(PQ, PV, PC)
    pick = lambda g, ep: min(K - 1, int(np.float32(rng_uniform(seed, g, ep, 1000)) * np.float32(K)))
    sim.reset()
    q, v, _ = sim.get_state()
    for e in range(n):
        k = pick(base + e, 0)
        np.testing.assert_array_equal(q[:, e].astype(np.float32), PQ[:, k])
        np.testing.assert_array_equal(v[:, e].astype(np.float32), PV[:, k])
    assert len({pick(base + e, 0) for e in range(n)}) > 1
    sim.step(np.zeros((n, 6), dtype=np.float32))
    obs, rew, disc, st = sim.step(np.zeros((n, 6), dtype=np.float32))
    assert np.all(st == 2)                                      # time limit
    obs, rew, disc, st = sim.step(np.zeros((n, 6), dtype=np.float32))
    assert np.all(st == 0)                                      # auto-reset -> FIRST, episode 1
    q, v, _ = sim.get_state()
    for e in range(n):
        k = pick(base + e, 1)
        np.testing.assert_array_equal(q[:, e].astype(np.float32), PQ[:, k])
        np.testing.assert_array_equal(obs[e, 12:18], PC[:, k])   # commanded pose = the pool's hold pose
    sim.set_reset_pool(None, None, None)
    sim.reset()
    q, _, _ = sim.get_state()
    assert np.all(q[:6] == 0)                                   # the reference reset again


def test_pickplace_pool_against_oracle(blobs):
    """The scripted pre-grasp pool (so101_sim_amd/pregrasp.py), built on the GPU through the product path:
    * grasp entries are contact-heavy and one control step from them matches the oracle (state tolerance of
      check_control_step with arm contact; reward / discount / step_type exact);
    * drop entries (banana released over the bowl) reach reward 1.0 / discount 0 / LAST on the GPU, and the oracle,
      started from the GPU state one step earlier, reports exactly the same (reward, discount, step_type)."""
    import torch
    from so101_sim_amd import pregrasp
    n = 64
    env = _batched_env("SO100HandOverBanana", n)
    PQ, PV, PC = (t.cpu().numpy().astype(np.float64) for t in pregrasp.build_pickplace_pool(env, pool_size=n, seed=3))
    env.close()
    half = n // 2
    # ---- forward pass at all 32 grasp entries + 8 drop entries
    # (a) narrowphase: the kernel's contact list against the oracle's (tolerances of _compare_contact_lists: depth and normal of
    #     EVERY contact).  With the default narrowphase (EPA: an exact face of the Minkowski difference) NO entry may differ -
    #     `narrow_differs == 0`; with the MPR option fp32 and fp64 ended on different portals on 12 / 3 / 6 of 32 entries (seeds
    #     3 / 4 / 5) where a thin finger pad touches a hull with an edge or a corner.  Only the witness point on a flat facet,
    #     which is not unique, may sit elsewhere (counted; at most 3 % of the contacts).
    # (b) solver: with the KERNEL's contact list injected into the oracle, constraint rows + Newton solve must agree to
    #     1e-3 of max|qacc| on every entry - stiff pad contacts (solimp clamped to 0.9999) squeezing a 70 g banana are
    #     the case in which a cost-based fp32 termination stops early (so101_newton.hpp, decrement test).
    grasp = list(range(half))
    simg = ArraySim(blobs["f32"], len(grasp), backend="gpu", last_step=500)
    simg.set_state(PQ[:, grasp], PV[:, grasp], PC[:, grasp], np.zeros((18, len(grasp))))
    dbgg = simg.debug_forward()
    #     Independently of the fp64 twin, the DEFINITION of a penetration depth (oracle/geomcheck.py, no shared algorithm) bounds
    #     every contacting pair of every grasp entry: overlap along its normal >= its depth >= the brute-forced minimum
    #     translation, and the depth IS the minimum translation (within 2 % for >= 99 % of the pairs, worst <= 1.10).
    from oracle import geomcheck as gc
    from so101_sim_amd.model import blob as blobfmt
    model = blobfmt.unpack(blobs["f64"])
    narrow_differs, rows, n_contacts, n_witness = 0, [], 0, 0
    for j, k in enumerate(grasp):
        o = Oracle(blobs["f64"])
        o.set_state(PQ[:, k], PV[:, k], np.zeros(18))
        o.set_ctrl(PC[:, k])
        o.forward()
        problems, t, l, w = pc._compare_contact_lists(dbgg[j]["contacts"], o.contacts())
        narrow_differs += bool(problems) or l > 0
        n_contacts += t; n_witness += w
        rows += gc.check_contacts(gc.Scene.from_oracle(model, o), dbgg[j]["contacts"])
    assert narrow_differs == 0, narrow_differs
    assert n_contacts >= 300 and n_witness <= 0.03 * n_contacts, (n_witness, n_contacts)
    mini = np.array([r["minimality"] for r in rows])
    deep = np.array([r["depth"] for r in rows]) > 1e-4        # (relative numbers only where the depth is above the arithmetic's position noise)
    assert len(rows) >= 200 and min(r["along"] - r["depth"] for r in rows) >= -5e-6 and min(r["depth"] - r["mtd"] for r in rows) >= -5e-6 - 2e-3 * max(r["mtd"] for r in rows)
    assert np.median(mini) <= 1.002 and np.mean(mini[deep] <= 1.02) >= 0.99 and mini[deep].max() <= 1.10, (np.median(mini), np.mean(mini[deep] <= 1.02), mini[deep].max())
    idx = list(range(0, 8)) + list(range(half, half + 8))
    sim = ArraySim(blobs["f32"], len(idx), backend="gpu", last_step=500)
    sim.set_state(PQ[:, idx], PV[:, idx], PC[:, idx], np.zeros((18, len(idx))))
    dbg = sim.debug_forward()
    ncon = []
    for j, k in enumerate(idx):
        o = Oracle(blobs["f64"])
        o.set_state(PQ[:, k], PV[:, k], np.zeros(18))
        o.set_ctrl(PC[:, k])
        o.forward()
        problems, _, l, _ = pc._compare_contact_lists(dbg[j]["contacts"], o.contacts())
        assert not problems and l == 0, (j, problems, l)
        ncon.append(len(o.contacts()))
        o.inject_contacts(dbg[j]["contacts"])
        o.forward()
        a = o.qacc()[0]
        err = np.abs(dbg[j]["qacc"] - a).max() / np.abs(a).max()
        assert err <= 1e-3, (j, err, dbg[j]["iters"])
    assert np.mean(ncon[:8]) >= 11, ncon                            # contact-heavy: banana + bowl on the table + the gripper (measured 11.9)
    # ---- one control step (ten substeps, jaws closing) from ALL 32 grasp entries and 8 drop entries: task outputs exact.
    # Drop entries: states to the free-space bound (arm parked), contact-phase bound when the banana already touches the rim.
    # Grasp entries: ten substeps of stiff pad contacts amplify any difference of the first substep, so the bound is a
    # distribution: the typical entry agrees like any contact phase (median <= 2e-4 rad / 1e-2 rad/s), at least 90 % are inside
    # 2e-3 rad / 0.1 rad/s, and the worst stays inside 2e-2 rad / 1 rad/s (round 3, MPR on both sides: 69-75 % and 0.033 / 9.1,
    # held by a bound of 5e-2 / 20 that checked nothing; EPA, round 3: 29 of 32, worst 5.7e-3 / 0.25 - entries whose witness
    # point sits elsewhere on a flat facet).
    idx2 = grasp + list(range(half, half + 8))
    sim = ArraySim(blobs["f32"], len(idx2), backend="gpu", last_step=500)
    sim.set_state(PQ[:, idx2], PV[:, idx2], PC[:, idx2], np.zeros((18, len(idx2))))
    sim.begin_episode()
    act = PC[:, idx2].T.astype(np.float32).copy()
    act[:, 5] -= 0.3
    obs, rew, disc, st = sim.step(act)
    q1, v1, _ = sim.get_state()
    gq, gv = [], []
    for j, k in enumerate(idx2):
        o = Oracle(blobs["f64"])
        o.env_config(seed=0, env_id=j, last_step=500)
        o.set_state(PQ[:, k], PV[:, k], np.zeros(18))
        o.env_begin()
        oo, orew, odisc, ost = o.env_step(act[j].astype(np.float64))
        qo, vo, _ = o.get_state()
        assert (rew[j], disc[j], st[j]) == (orew, odisc, ost), j
        dq, dv = np.abs(q1[:, j] - qo).max(), np.abs(v1[:, j] - vo).max()
        if j < len(grasp):
            gq.append(dq); gv.append(dv)
            assert dq <= 2e-2 and dv <= 1.0, (j, dq, dv)
        else:
            touching = ncon[8 + j - len(grasp)] > 13
            tq, tv = (2e-3, 0.1) if touching else (2e-5, 5e-3)
            assert dq <= tq and dv <= tv, (j, dq, dv)
    gq, gv = np.array(gq), np.array(gv)
    assert np.median(gq) <= 2e-4 and np.median(gv) <= 1e-2, (np.median(gq), np.median(gv))
    assert np.mean((gq <= 2e-3) & (gv <= 0.1)) >= 0.9, (np.sort(gq)[-12:], np.sort(gv)[-12:])
    # ---- jaws squeezing a banana of randomised mass: GPU vs oracle with the same scale, and the scale matters
    g8 = list(range(8))
    scale = np.random.RandomState(2).uniform(0.5, 1.5, size=(2, 8))
    sim = ArraySim(blobs["f32"], 8, backend="gpu", last_step=500)
    sim.set_mass_scale(scale)
    sim.set_state(PQ[:, g8], PV[:, g8], PC[:, g8], np.zeros((18, 8)))
    sim.begin_episode()
    sim.step(act[:8])
    qs, vs, _ = sim.get_state()
    changed = 0.0
    for j in g8:
        o = Oracle(blobs["f64"])
        o.env_config(seed=0, env_id=j, last_step=500)
        o.set_mass_scale(scale[:, j])
        o.set_state(PQ[:, j], PV[:, j], np.zeros(18))
        o.env_begin()
        o.env_step(act[j].astype(np.float64))
        qo, vo, _ = o.get_state()
        assert np.abs(qs[:, j] - qo).max() <= 2e-2 and np.abs(vs[:, j] - vo).max() <= 1.0, (j, np.abs(qs[:, j] - qo).max(), np.abs(vs[:, j] - vo).max())
        changed = max(changed, np.abs(vs[:, j] - v1[:, j]).max())
    assert changed > 1e-3, "the mass scale must change how the gripper moves the banana"
    # ---- the reward = 1 branch: drop entries until the banana rests in the bowl
    m = 16
    sim = ArraySim(blobs["f32"], m, backend="gpu", last_step=500, settle_max_substeps=50)   # (finished envs auto-reset cheaply)
    sim.set_state(PQ[:, half:half + m], PV[:, half:half + m], PC[:, half:half + m], np.zeros((18, m)))
    sim.begin_episode()
    hold = PC[:, half:half + m].T.astype(np.float32).copy()
    hits = 0
    for t in range(120):
        before = [a.copy() for a in sim.get_state()]
        obs, rew, disc, st = sim.step(hold)
        for e in np.nonzero(rew == 1.0)[0]:
            assert disc[e] == 0.0 and st[e] == 2
            o = Oracle(blobs["f64"])
            o.env_config(seed=0, env_id=int(e), last_step=500)
            o.set_state(before[0][:, e], before[1][:, e], before[2][:, e])
            o.env_begin()
            _, orew, odisc, ost = o.env_step(hold[e].astype(np.float64))
            assert (orew, odisc, ost) == (1.0, 0.0, 2), (t, e, orew, odisc, ost)
            hits += 1
        if hits >= 3:
            break
    assert hits >= 1, "no env reached reward 1.0 from the drop entries"


def test_pickplace_properties_at_16384():
    """configs[2] at full size with the benchmarked solver settings: 16384 envs from the pre-grasp pool, hold pose +
    N(0, 0.05) actions with the jaw closing.  Rewards in {0,1}, success <=> discount 0 <=> LAST (no time limit hit),
    finite state, mean contacts per env >= 8, some env reaches reward 1, ep_return = sum of rewards, overflow rare."""
    import torch
    from so101_sim_amd import pregrasp
    n = 16384
    env = _batched_env("SO100HandOverBanana", n)
    pool = pregrasp.build_pickplace_pool(env, pool_size=4096, seed=0)
    env.set_reset_pool(*pool)
    env.reset_all()
    env.events(clear=True)
    gen = torch.Generator(device=env.device)
    gen.manual_seed(0)
    hold = env.obs[:, 12:18].clone()
    total = torch.zeros(n, device=env.device)
    successes, ncon_sum = 0, 0.0
    steps = 40
    for t in range(steps):
        first = (env.step_type == 0).unsqueeze(1)
        hold = torch.where(first, env.obs[:, 12:18], hold)
        noise = 0.05 * torch.randn(n, 6, device=env.device, generator=gen)
        act = hold + noise
        act[:, 5] = hold[:, 5] - 0.3 + noise[:, 5]
        obs, rew, disc, st = env.step_tensor(act)
        assert bool(((rew == 0) | (rew == 1)).all())
        mid = st != 0
        assert bool((disc[mid & (rew == 1)] == 0).all()) and bool((st[rew == 1] == 2).all())
        total = torch.where(st == 0, torch.zeros_like(total), total + rew)
        successes += int((rew == 1).sum())
        ncon_sum += float(env.diagnostics()[:, 0].float().mean())
    assert bool(torch.isfinite(env.qpos).all()) and bool(torch.isfinite(env.qvel).all())
    assert successes >= 10, successes
    assert ncon_sum / steps >= 8.0, ncon_sum / steps
    np.testing.assert_array_equal(env.episode_returns().cpu().numpy(), total.cpu().numpy())
    ev = env.events()
    env_steps = n * steps
    assert ev["contact_overflow"] <= 0.01 * env_steps and ev["candidate_overflow"] <= 0.01 * env_steps, ev
    assert ev["diverged"] <= 0.01 * env_steps, ev
    env.close()


# ------------------------------------------------------------------ mixed suite (configs[3]) and the configs[4] share
def test_mixed_suite_two_streams():
    """configs[3]: Banana and Pen handles side by side on two streams with per-env mass scales; each handle's result is
    bit-identical to running it alone (the handles share nothing), rewards / discounts stay in range."""
    import torch
    n = 2048
    acts = {}
    def run(names, concurrent):
        out = {}
        envs = {nm: _batched_env(nm, n, settle_max_substeps=200) for nm in names}
        gen = torch.Generator(device="cuda")
        streams = {nm: (torch.cuda.Stream() if concurrent else torch.cuda.current_stream()) for nm in names}
        for nm, env in envs.items():
            gen.manual_seed(7 + len(nm))
            env.set_mass_scale(0.5 + torch.rand(2, n, device=env.device, generator=gen))
            acts.setdefault(nm, torch.rand(6, n, 6, device=env.device, generator=gen) * 2 - 1)
        torch.cuda.synchronize()
        for nm, env in envs.items():
            with torch.cuda.stream(streams[nm]):
                env.reset_all()
        for t in range(6):
            for nm, env in envs.items():
                with torch.cuda.stream(streams[nm]):
                    env.step_tensor(acts[nm][t])
        torch.cuda.synchronize()
        for nm, env in envs.items():
            assert bool(torch.isfinite(env.qpos).all()) and bool(((env.reward == 0) | (env.reward == 1)).all())
            out[nm] = (env.qpos.clone(), env.qvel.clone(), env.obs.clone())
            env.close()
        return out
    both = run(["SO100HandOverBanana", "SO100HandOverPen"], True)
    for nm in both:
        alone = run([nm], False)
        for a, b in zip(both[nm], alone[nm]):
            assert torch.equal(a, b), nm


def test_mixed_suite_at_full_size_with_oracle_slices(blobs, blobs_pen):
    """configs[3] at its BASELINE.json size: 16384 x Banana + 16384 x Pen on two streams with per-env mass scales.  One control
    step from the reset state of a slice of envs of EACH handle is repeated by the fp64 oracle with the same mass scale
    (resting props, free arm: the free-space / resting-contact bound of check_control_step); properties on all 32768 envs."""
    import torch
    n, k = 16384, 6
    names = {"SO100HandOverBanana": blobs, "SO100HandOverPen": blobs_pen}
    envs = {nm: _batched_env(nm, n, settle_max_substeps=1000) for nm in names}
    streams = {nm: torch.cuda.Stream() for nm in names}
    gen = torch.Generator(device="cuda")
    act, before = {}, {}
    for nm, env in envs.items():
        with torch.cuda.stream(streams[nm]):
            gen.manual_seed(11 + len(nm))
            env.set_mass_scale(0.5 + torch.rand(2, n, device=env.device, generator=gen))
            act[nm] = 0.5 * (torch.rand(n, 6, device=env.device, generator=gen) * 2 - 1)
            env.reset_all()
            before[nm] = (env.qpos[:, :k].clone(), env.qvel[:, :k].clone(), env.warm[:, :k].clone(), env.ctrl[:, :k].clone())
    for nm, env in envs.items():
        with torch.cuda.stream(streams[nm]):
            env.step_tensor(act[nm])
    torch.cuda.synchronize()
    for nm, env in envs.items():
        assert bool(torch.isfinite(env.qpos).all()) and bool(torch.isfinite(env.qvel).all())
        assert bool(((env.reward == 0) | (env.reward == 1)).all()) and bool((env.step_type == 1).all())
        assert torch.allclose(env.qpos[9:13].norm(dim=0), torch.ones(n, device=env.device), atol=1e-5)
        ev = env.events()
        assert ev["diverged"] == 0 and ev["scheduler_abort"] == 0 and ev["candidate_overflow"] == 0, ev
        q0, v0, w0, c0 = (t.cpu().numpy().astype(np.float64) for t in before[nm])
        ms = env.mass_scale[:, :k].cpu().numpy().astype(np.float64)
        for e in range(k):
            o = Oracle(names[nm]["f64"])
            o.set_mass_scale(ms[:, e])
            o.set_state(q0[:, e], v0[:, e], w0[:, e])
            o.set_ctrl(act[nm][e].cpu().numpy().astype(np.float64))
            o.substeps(10)
            qo, vo, _ = o.get_state()
            dq = np.abs(env.qpos[:, e].cpu().numpy() - qo).max(); dv = np.abs(env.qvel[:, e].cpu().numpy() - vo).max()
            assert dq <= 2e-5 and dv <= 2e-3, (nm, e, dq, dv)
        env.close()


@pytest.mark.parametrize("seed", [1, 7, 3])
def test_failure_rates_on_the_headline_workload(seed):
    """4096 envs x 500 control steps of uniform random actions over the action spec (three seeds): how often the physics diverge
    (the episode then ends like a dm_control PhysicsError), how often contacts are DROPPED - round 5: never for arm-link contacts (the
    Jacobian pool's tail is recomputed, event 2 is retired), and for the contact list only when the touching geom pairs alone exceed 64; an
    env with more than 64 contacts keeps one contact per pair for that substep (event 7, `contacts_reduced`, bounded here) -, and that
    no candidate list overflows and the scheduler never aborts.  Bounds: 5e-5 divergences per env-step (measured with the default
    narrowphase: 2.0e-5; the MPR option: 7.2e-5; DESIGN.md section 4 on why uniform random actions do that).  At five points of the rollout
    one control step of 24 envs is repeated by the fp64 oracle from the same live state (one-step parity on rollout states)."""
    import torch
    n, steps = 4096, 500
    env = _batched_env("SO100HandOverBanana", n)
    spec = env.action_spec()
    lo = torch.tensor(spec.minimum, device=env.device); hi = torch.tensor(spec.maximum, device=env.device)
    gen = torch.Generator(device=env.device); gen.manual_seed(seed)
    st = torch.cuda.Stream()
    from so101_sim_amd.model import scenes
    raw64, _ = scenes.load_blob("banana", "f64")
    k, checks = 24, []
    with torch.cuda.stream(st):
        env.reset_all()
        env.events(clear=True)
        for t in range(steps):
            act = lo + (hi - lo) * torch.rand(n, 6, device=env.device, generator=gen)
            probe = t in (60, 180, 300, 420, 480)
            if probe:        # the state these envs start the step from, for the one-step comparison below
                before = [x[:, :k].clone() for x in (env.qpos, env.qvel, env.warm)]
            env.step_tensor(act)
            if probe:
                checks.append((before, act[:k].clone(), env.qpos[:, :k].clone(), env.qvel[:, :k].clone(), env.step_type[:k].clone()))
    torch.cuda.synchronize()
    # one control step of 5 x 24 envs at five points of the rollout (live states: props pushed around, arm on the table)
    # repeated by the fp64 oracle from the same state
    err = []
    for before, act, q1, v1, stp in checks:
        b = [x.cpu().numpy().astype(np.float64) for x in before]
        for e in range(k):
            if int(stp[e]) != 1:
                continue                                   # the env reset in this call
            o = Oracle(raw64)
            o.set_state(b[0][:, e], b[1][:, e], b[2][:, e])
            o.set_ctrl(act[e].cpu().numpy().astype(np.float64))
            o.substeps(10)
            qo, vo, _ = o.get_state()
            err.append((np.abs(q1[:, e].cpu().numpy() - qo).max(), np.abs(v1[:, e].cpu().numpy() - vo).max()))
    err = np.array(err)
    # bar: >= 95 % inside 2e-3 rad / 0.1 rad/s, worst <= 2e-2 rad / 3 rad/s (round 5: 5 rad/s; round 3 under MPR: median 7e-7 / 4e-5, 90 %, worst 0.012 / 1.8)
    assert len(err) >= 100
    assert np.median(err[:, 0]) <= 2e-5 and np.median(err[:, 1]) <= 2e-3, (np.median(err[:, 0]), np.median(err[:, 1]))
    # (seed 1: 99 % inside, worst 1.3e-3 rad / 0.18 rad/s; seed 7, added in round 5: 98 %, worst 2.1e-3 rad / 2.6 rad/s - one probe with the arm pressing a prop)
    assert np.mean((err[:, 0] <= 2e-3) & (err[:, 1] <= 0.1)) >= 0.95 and err[:, 0].max() <= 2e-2 and err[:, 1].max() <= 3.0, (
        np.mean((err[:, 0] <= 2e-3) & (err[:, 1] <= 0.1)), err[:, 0].max(), err[:, 1].max())
    ev = env.events()
    per = {k: v / (n * steps) for k, v in ev.items()}
    assert per["diverged"] <= 5e-5, ev
    # no contact is dropped: arm-link contacts beyond the 40-slot LDS pool recompute their Jacobian (event 2 stays 0; round 4 dropped them on
    # 1.8e-4 - 4.6e-4 of the env-steps), and the contact list is only cut when more than 64 geom PAIRS touch (<= 1e-5);
    # more than 64 CONTACTS - the arm jammed into the bowl's 53 hulls, hull pairs carrying patches since round 5 - reduce the env to one
    # contact per pair for that substep
    assert per["contact_overflow"] <= 1e-5 and ev["arm_pool_overflow"] == 0 and per["contacts_reduced"] <= 2e-4, ev
    assert ev["candidate_overflow"] == 0 and ev["scheduler_abort"] == 0 and ev["placement_rejected"] == 0, ev
    assert bool(torch.isfinite(env.qpos).all())
    env.close()


def test_properties_at_32768_envs(make_sim):
    """Per-GPU share of configs[4] (262144 envs over 8 GPUs) with the benchmarked solver settings (100 iterations,
    tolerance 1e-8): reset + 5 random-action steps; finite state, unit quaternions, arm untouched by the reset,
    rewards in {0,1}, commanded pose = action, ep_return = sum of rewards, overflow / divergence events rare."""
    n = 32768
    sim = make_sim(n, seed=2, last_step=500)
    sim.reset()
    q, v, _ = sim.get_state()
    assert np.all(np.isfinite(q)) and np.all(np.isfinite(v)) and np.all(q[:6] == 0)
    np.testing.assert_allclose(np.linalg.norm(q[9:13], axis=0), 1.0, atol=1e-5)
    np.testing.assert_allclose(np.linalg.norm(q[16:20], axis=0), 1.0, atol=1e-5)
    sim.get_events(clear=True)
    rng = np.random.RandomState(3)
    total = np.zeros(n)
    steps = 5
    for t in range(steps):
        act = rng.uniform(LO, HI, size=(n, 6)).astype(np.float32)
        obs, rew, disc, st = sim.step(act)
        total += rew
        assert set(np.unique(rew)) <= {0.0, 1.0} and set(np.unique(st)) <= {1, 2}
        np.testing.assert_array_equal(obs[:, 12:18], act)
    q, v, _ = sim.get_state()
    assert np.all(np.isfinite(q)) and np.all(np.isfinite(v))
    np.testing.assert_array_equal(sim._get(sim.ep_return), total.astype(np.float32))
    ev = sim.get_events()
    assert ev["contact_overflow"] + ev["candidate_overflow"] + ev["arm_pool_overflow"] + ev["contacts_reduced"] <= 1e-3 * n * steps, ev
    assert ev["diverged"] <= 1e-3 * n * steps, ev


def test_env_on_a_non_current_device_guard():
    """Every C-ABI entry point makes the handle's device current: with one visible GPU the guard is a no-op,
    so this checks the call sequence under a changed *stream* context and that create/destroy leave the device alone."""
    import torch
    dev_before = torch.cuda.current_device()
    env = _batched_env("SO100HandOverBanana", 8, settle_max_substeps=50)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        env.reset_all()
        env.step_tensor(torch.zeros(8, 6, device=env.device))
    torch.cuda.synchronize()
    assert torch.cuda.current_device() == dev_before and bool(torch.isfinite(env.qpos).all())
    ev = env.events()
    assert set(ev) == {"candidate_overflow", "contact_overflow", "arm_pool_overflow", "diverged", "placement_rejected", "settle_not_converged",
                       "scheduler_abort", "contacts_reduced"} and ev["scheduler_abort"] == 0
    env.close()


def test_batched_physics_state_and_its_15_step_delay():
    """physics_state = concat(qpos, qvel) and delayed_physics_state (15 control steps, padded with the episode's first value:
    so100_task.py:203-210,366-368) for a batch, across a time-limit auto-reset, against a host-side replay of the same line."""
    import collections
    import torch
    env = _batched_env("SO100HandOverBanana", 6, settle_max_substeps=40, time_limit=0.4, physics_state=True)    # 20-step episodes
    ts = env.reset()
    assert list(ts.observation)[:4] == ["commanded_joints_pos", "joints_pos", "joints_vel", "physics_state"] and list(ts.observation)[-1] == "delayed_physics_state"
    assert ts.observation["physics_state"].shape == (6, 38)
    state = lambda: torch.cat([env.qpos, env.qvel]).t().clone()
    lines = [collections.deque([state()[e]] * 15, maxlen=15) for e in range(6)]
    assert torch.equal(env.physics_state, state()) and torch.equal(env.delayed_physics_state, state())
    g = torch.Generator(device=env.device); g.manual_seed(3)
    firsts = 0
    for t in range(47):
        ts = env.step(0.3 * torch.randn(6, 6, device=env.device, generator=g))
        cur = state()
        assert torch.equal(env.physics_state, cur)
        for e in range(6):
            if int(env.step_type[e]) == 0:                       # auto-reset: the line restarts from the new episode's first state
                lines[e] = collections.deque([cur[e]] * 15, maxlen=15); firsts += 1
            expect = lines[e][0]
            lines[e].append(cur[e])
            assert torch.equal(env.delayed_physics_state[e], expect), (t, e)
    assert firsts >= 6
    env.close()


def test_single_env_orders_its_own_stream_behind_the_callers():
    """SingleEnvironment steps on a private non-blocking stream; a mutator enqueued on the CALLER's stream right before reset()
    must be seen by that reset.  Here the reset pool (the state an episode starts from) only comes into existence on a side stream behind
    tens of milliseconds of matrix products; reset() is called at once, from that stream.  Without the stream edge the reset kernel reads
    the pool's memory before it is written."""
    import os
    import torch
    from so101_sim_amd import task_suite
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        env = task_suite.create_task_env("SO100HandOverBanana", time_limit=10.0, random_state=5)
    finally:
        os.chdir(cwd)
    dev = env.device
    ts = env.reset()
    q = torch.tensor(ts.observation["physics_state"][:20], dtype=torch.float32, device=dev)
    arm = torch.tensor([0.1, -1.2, 1.3, 0.4, -0.2, 0.03], device=dev)
    q[:6] = arm
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    for trial in range(3):
        with torch.cuda.stream(side):
            big = torch.randn(4096, 4096, device=dev)
            for _ in range(20):
                big = (big @ big) * 1e-3
            late = 0.0 * big[0, :1].nan_to_num()                  # (exists only once the products are done)
            qpos = torch.full((20, 1), 7.0, device=dev)           # fresh memory, wrong contents until the line below has run
            qpos.copy_((q + late).unsqueeze(1))
            env.set_reset_pool(qpos, torch.zeros(18, 1, device=dev), arm.unsqueeze(1).clone())
            ts = env.reset()
        np.testing.assert_allclose(ts.observation["joints_pos"], arm.cpu().numpy(), atol=1e-6, err_msg=f"trial {trial}")
        env.set_reset_pool(None)
    env.close()


def test_single_env_reset_is_seed_compatible_with_the_reference(blobs):
    """create_task_env(..., random_state=123) (the seed of the reference's own test, hand_over_test.py:34-39): the
    placements are numpy RandomState(123) draws in dm_control's PropPlacer order - object xyz (3), object yaw (1),
    container xyz (3 per attempt) - and the settled state matches the oracle settling the same placements; the next
    episode continues the same stream."""
    import os
    from so101_sim_amd import task_suite
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        env = task_suite.create_task_env("SO100HandOverBanana", time_limit=0.06, random_state=123)
    finally:
        os.chdir(cwd)
    ts = env.reset()
    rs = np.random.RandomState(123)
    opos = rs.uniform([0.2, -0.1, 0.45], [0.3, 0.1, 0.45])
    yaw = rs.uniform(-np.pi * 0.1, np.pi * 0.1)
    cpos = rs.uniform([-0.3, -0.1, 0.45], [-0.2, 0.1, 0.45])
    np.testing.assert_array_equal(env.placements["object_position"], opos)
    assert env.placements["object_yaw"] == yaw
    np.testing.assert_array_equal(env.placements["container_position"], cpos)      # first attempt accepted for this seed
    assert ts.first() and np.all(ts.observation["joints_pos"] == 0)
    state = ts.observation["physics_state"]
    q = state[:20]
    assert abs(q[6] - opos[0]) < 5e-3 and abs(q[7] - opos[1]) < 5e-3 and abs(q[13] - cpos[0]) < 5e-3 and abs(q[14] - cpos[1]) < 5e-3
    assert 0.4210 < q[8] < 0.4225 and abs(2 * np.arctan2(q[12], q[9]) - yaw) < 2e-2          # banana came to rest, yaw kept
    # the oracle, same placements, same settle rule
    o = Oracle(blobs["f64"])
    q0 = np.zeros(20)
    q0[6:9] = opos
    q0[9:13] = [np.cos(0.5 * yaw), 0, 0, np.sin(0.5 * yaw)]
    q0[13:16] = cpos
    q0[16] = 1.0
    o.set_state(q0, np.zeros(18), np.zeros(18))
    o.set_ctrl(np.array([0.0, -1.57079, 1.57079, 1.57079, -1.57079, 0.0]))
    for k in range(1000):
        o.substeps(1, True)
        _, v, _ = o.get_state()
        if np.abs(v[6:]).max() < 1e-3 and o.L.orc_max_prop_qacc(o.h) < 1e-2:
            break
    qo, _, _ = o.get_state()
    assert np.abs(q - qo).max() < 5e-3, np.abs(q - qo).max()
    # three steps to the time limit, then the next reset draws the following numbers of the SAME stream
    for _ in range(3):
        ts = env.step(np.zeros(6))
    assert ts.last()
    ts = env.step(np.zeros(6))
    assert ts.first()
    opos2 = rs.uniform([0.2, -0.1, 0.45], [0.3, 0.1, 0.45])
    np.testing.assert_array_equal(env.placements["object_position"], opos2)
    env.close()


def test_settled_cache_file_roundtrip(tmp_path):
    """the settled states of the first episodes are written to disk by one environment and picked up by a
    second one built the same way; its resets then copy them (bit-identical to an environment that settles), and a file
    made for another seed or another mass scale is refused."""
    import torch
    from so101_sim_amd import task_suite, settled_cache
    N, path = 128, str(tmp_path / "settled.bin")
    mk = lambda seed=3: task_suite.create_task_env("SO100HandOverBanana", time_limit=0.1, n_envs=N, random_state=seed,
                                                   device="cuda:0", prefetch_resets=False)
    act = torch.zeros(N, 6, device="cuda:0")

    def rollout(env, steps=13):
        env.reset()
        out = [torch.cat([env.qpos, env.qvel]).cpu().numpy().copy()]
        for _ in range(steps):
            env.step_tensor(act)
            out.append(torch.cat([env.qpos, env.qvel, env.obs.T, env.reward[None], env.discount[None]]).cpu().numpy().copy())
        return out, env.events()

    a = mk()
    ref, ev_ref = rollout(a)                           # settles every reset
    a.close()
    b = mk()
    b.save_settled_cache(path, n_episodes=3)
    b.close()
    c = mk()
    header = c.load_settled_cache(path)
    assert header["n_episodes"] == 3 and header["n_envs"] == N and header["first_episode"] == 0
    got, ev_got = rollout(c)
    assert int(c.episode.min()) >= 3                   # 13 steps at a 5-step limit: episodes 0, 1, 2 were all started
    for x, y in zip(ref, got):
        np.testing.assert_array_equal(x, y)
    assert ev_ref == ev_got
    c.close()
    d = mk(seed=4)
    with pytest.raises(settled_cache.SettledCacheError, match="seed"):
        d.load_settled_cache(path)
    d.close()
    e = mk()
    e.set_mass_scale(torch.full((2, N), 1.1))
    with pytest.raises(settled_cache.SettledCacheError, match="mass_scale_sha256"):
        e.load_settled_cache(path)
    e.close()


def test_config0_single_env_500_random_steps(blobs):
    """BASELINE.json configs[0]: `create_task_env('SO100HandOverBanana', time_limit=10.0, random_state=0)`, ONE env, 500
    uniform random actions through the reference's own Python surface.  The oracle's env layer, started from the same
    reset state, runs beside it for the first steps (arm joints to 1e-4 until the arm touches something, 5e-3 for five more
    steps): exact task outputs and the observation delay line throughout; the episode must end with LAST on control step 500 (or earlier on a physics error /
    success, discount 0) and auto-reset.
    ALL 500 steps are also checked one by one: a second oracle is put on the state the step started from (qpos, qvel, warm start)
    and makes the same ten substeps - free trajectories separate chaotically once the arm hits something, single steps do not.
    Without arm contact (147 steps): EVERY step during which the set of touching pairs does not change inside 2e-5 rad / 2e-2 rad/s (measured
    8e-6 / 4e-3), the steps in which a pair starts or stops touching - detected from the oracle's substeps (round 6): about a third of them, marginal
    pieces of a resting prop at depth ~ 0; the one outlier of round 5 is among them: 1.1e-4 / 0.09, its ten substeps agree to 4e-5 rad/s each when re-started from the kernel's state, scripts/gpu_config0_debug.py -
    inside 5e-4 / 0.5; with arm contact: median below 1e-5 rad (3e-7), at least 97 % inside
    2e-3 rad / 0.1 rad/s (99.8 %), every step inside 2e-2 / 1 (1.3e-3 / 0.18; the bound was 5e-2 / 20 until round 4)."""
    from so101_sim_amd import task_suite
    cwd = os.getcwd()
    os.chdir(os.path.dirname(os.path.abspath(__file__)))       # no calibration/red_arm.json here: offsets are zero
    try:
        env = task_suite.create_task_env("SO100HandOverBanana", time_limit=10.0, random_state=0)
    finally:
        os.chdir(cwd)
    spec = env.action_spec()
    ts = env.reset()
    assert ts.first() and ts.reward is None and ts.discount is None
    assert list(ts.observation) == ["commanded_joints_pos", "joints_pos", "joints_vel", "physics_state", "undelayed_joints_pos",
                                    "undelayed_joints_vel", "delayed_physics_state"]
    state0 = np.asarray(ts.observation["physics_state"], dtype=np.float64)
    o = Oracle(blobs["f64"])
    o.env_config(last_step=500)
    o.set_state(state0[:20], state0[20:], np.zeros(18))
    o.set_ctrl(np.asarray(ts.observation["commanded_joints_pos"], dtype=np.float64))
    o.env_begin()
    rng = np.random.RandomState(0)
    undelayed, compared, touched, budget, ended = [], 0, False, 5, None
    o1 = Oracle(blobs["f64"])                 # the one-step checker
    one_step = []
    for t in range(1, 501):
        a = rng.uniform(spec.minimum, spec.maximum).astype(np.float32)
        before = [x[:, 0].cpu().numpy().astype(np.float64) for x in (env.qpos, env.qvel, env.warm)]
        ts = env.step(a)
        ob = ts.observation
        if not ts.last():
            o1.set_state(*before)
            o1.set_ctrl(a.astype(np.float64))
            # substep by substep: a step during which the SET of touching pairs changes (a prop-on-table contact that opens and closes) is a
            # discrete event that fp32 and fp64 may take one substep apart - told apart here instead of hidden in a quantile
            sigs, arm = [], False
            for _ in range(10):
                o1.substeps(1)
                cs = o1.contacts()
                sigs.append(tuple(sorted(set((c["geom1"], c["geom2"]) for c in cs))))
                arm = arm or any(pc._arm_geom(c["geom1"]) or pc._arm_geom(c["geom2"]) for c in cs)
            q1, v1, _ = o1.get_state()
            one_step.append((np.abs(ob["physics_state"][:20] - q1).max(), np.abs(ob["physics_state"][20:] - v1).max(), arm, len(set(sigs)) > 1))
        assert ob["joints_pos"].shape == (6,) and ob["joints_vel"].shape == (0,) and ob["physics_state"].shape == (38,)
        assert np.all(np.isfinite(ob["physics_state"])) and ts.reward in (0.0, 1.0)
        np.testing.assert_array_equal(ob["commanded_joints_pos"], a.astype(np.float64))        # unclamped ctrl, zero offsets
        np.testing.assert_array_equal(ob["undelayed_joints_pos"], ob["physics_state"][:6])
        undelayed.append(ob["undelayed_joints_pos"].copy())
        want = undelayed[t - 6] if t >= 6 else np.zeros(6)                                   # value of control step t - 5
        np.testing.assert_array_equal(ob["joints_pos"], want)
        if budget > 0 and not ts.last():
            # full-range random targets drive the arm into the table or itself within a step or two; from then on the
            # trajectories separate chaotically, so the comparison loosens and stops five steps later
            oo, orew, odisc, ost = o.env_step(a.astype(np.float64))
            touched = touched or any(pc._arm_geom(c["geom1"]) or pc._arm_geom(c["geom2"]) for c in o.contacts())
            np.testing.assert_allclose(ob["undelayed_joints_pos"], oo[6:12], atol=5e-3 if touched else 1e-4)
            assert (float(ts.reward), float(ts.discount)) == (orew, odisc)
            compared += 1
            budget -= 1 if touched else 0
        if ts.last():
            ended = t
            break
        assert ts.mid() and ts.discount == 1.0
    assert compared >= 5, compared
    r = np.array(one_step, dtype=np.float64)
    free, arm = r[r[:, 2] == 0], r[r[:, 2] == 1]
    assert len(r) >= 400 and len(arm) >= 100, (len(r), len(arm))
    # without arm contact: EVERY step whose set of touching pairs stays what it is inside 2e-5 rad / 2e-2 rad/s (the hard bound of round 4); the
    # steps in which a pair starts or stops touching (round 5: one of 147) inside 5e-4 / 0.5
    steady, switching = free[free[:, 3] == 0], free[free[:, 3] == 1]
    # (a resting prop's marginal pieces touch at depth ~ 0 and come and go: about a third of the free steps see the set change somewhere)
    assert len(steady) >= 0.5 * len(free), (len(steady), len(free))
    assert steady[:, 0].max() <= 2e-5 and steady[:, 1].max() <= 2e-2, (steady[:, 0].max(), steady[:, 1].max(), len(steady))
    assert len(switching) == 0 or (switching[:, 0].max() <= 5e-4 and switching[:, 1].max() <= 0.5), (switching[:, 0].max(), switching[:, 1].max(), len(switching))
    assert np.mean((free[:, 0] <= 2e-5) & (free[:, 1] <= 2e-2)) >= 0.99, np.mean((free[:, 0] <= 2e-5) & (free[:, 1] <= 2e-2))
    assert np.median(arm[:, 0]) <= 1e-5 and arm[:, 0].max() <= 2e-2 and arm[:, 1].max() <= 1.0, (np.median(arm[:, 0]), arm[:, 0].max(), arm[:, 1].max())
    assert np.mean((arm[:, 0] <= 2e-3) & (arm[:, 1] <= 0.1)) >= 0.97, np.mean((arm[:, 0] <= 2e-3) & (arm[:, 1] <= 0.1))
    assert ended is not None and (ended == 500 or ts.discount == 0.0), (ended, ts.discount)
    if ended == 500:
        assert ts.discount == 1.0 and ts.reward == 0.0
    assert env.step(np.zeros(6, dtype=np.float32)).first()
    env.close()
