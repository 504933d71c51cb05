"""MI355X parity tests: the HIP path, called through the C ABI, against the fp64 oracle on the same
seeded inputs, plus size-independent properties at the benchmark size (4096 envs)."""
import numpy as np
import pytest

from tests import parity_cases as pc
from tests.simharness import ArraySim

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def make_sim(blobs):
    def f(n, seed=0, **cfg):
        return ArraySim(blobs["f32"], n, backend="gpu", seed=seed, **cfg)
    return f


def test_native_library_is_the_hip_build():
    import os
    from so101_sim_amd import native
    assert os.path.exists(native.LIB_PATH)
    assert native.load_library().so101_version() == 10


def test_forward_stages(make_sim, blobs):
    pc.check_forward_stages(make_sim, blobs, n=16)


def test_pgs_forward_matches_oracle_pgs(make_sim, blobs):
    pc.check_pgs_forward(make_sim, blobs, n=8, iterations=100)


def test_pgs_control_step_runs_the_fused_path(make_sim, blobs):
    """so101_step with the PGS solver (fused kernel, no pipeline / prefetch): one control step against the oracle's PGS
    with the same sweep count; task outputs exact, arm state to the PGS-iterate tolerance."""
    from so101_sim_amd import native
    n = 4
    Q, V = pc.valid_arm_states(blobs["f64"], n, seed=6)
    rng = np.random.RandomState(7)
    act = rng.uniform(-0.5, 0.5, size=(n, 6)).astype(np.float32)
    sim = make_sim(n, solver=native.SOLVER_PGS, solver_iterations=50, solver_tolerance=0.0, last_step=500)
    sim.set_state(Q, V, np.zeros((6, n)), np.zeros((18, n)))
    sim.begin_episode()
    obs, rew, disc, st = sim.step(act)
    q1, v1, _ = sim.get_state()
    from oracle.oracle import Oracle
    for e in range(n):
        o = Oracle(blobs["f64"])
        o.set_solver_type(False)
        o.set_solver(50, 0.0)
        o.env_config(seed=0, env_id=e, last_step=500)
        o.set_state(Q[:, e], V[:, e], np.zeros(18))
        o.env_begin()
        _, orew, odisc, ost = o.env_step(act[e].astype(np.float64))
        qo, vo, _ = o.get_state()
        assert (rew[e], disc[e], st[e]) == (orew, odisc, ost)
        assert np.abs(q1[:6, e] - qo[:6]).max() <= 2e-4 and np.abs(v1[:6, e] - vo[:6]).max() <= 5e-2, (e, np.abs(q1[:6, e] - qo[:6]).max(), np.abs(v1[:6, e] - vo[:6]).max())


def test_kat1_through_the_kernels(make_sim, blobs, golden):
    pc.check_kat1(make_sim, blobs, golden)


def test_one_control_step(make_sim, blobs):
    pc.check_control_step(make_sim, blobs, n=16, iterations=100)


def test_reward_bitexact(make_sim, blobs):
    pc.check_reward_bitexact(make_sim, blobs, n=512)


def test_env_semantics(make_sim, blobs):
    pc.check_env_semantics(make_sim, blobs, n=4, settle=200, steps=9, last_step=7, iterations=50)


def test_reset_prefetch_is_bit_identical(make_sim):
    pc.check_prefetch_identical(make_sim, n=64, settle=300, steps=14, last_step=3)


@pytest.mark.parametrize("prefetch", [0, 1])
def test_settled_store_is_bit_identical(make_sim, prefetch):
    pc.check_settled_store_identical(make_sim, n=64, settle=300, steps=14, last_step=3, first=1, count=2, prefetch=prefetch)


def test_pipelined_step_matches_fused(make_sim, golden):
    pc.check_pipeline_identical(make_sim, golden, n=8, steps=6)


@pytest.mark.parametrize("n,steps", [(8, 6), (512, 4)])
def test_row_pass_of_the_narrowphase_matches_fused(make_sim, golden, monkeypatch, n, steps):
    """k_narrow<ROWS> - the instance large batches run: four light pairs per wavefront, one per DPP row, closed forms only, the rest by the
    whole wavefront - forced on for a small batch (SO101_NARROW_ROWS is read when a handle enqueues its step): against the fused step,
    bit for bit, on the contact-rich states."""
    monkeypatch.setenv("SO101_NARROW_ROWS", "1")
    pc.check_pipeline_identical(make_sim, golden, n=n, steps=steps, seed=11, all_reset_last=(n == 8), pipelines=(0, 1))


def test_four_launch_chains_match_fused_at_512_envs(make_sim, golden):
    """512 envs: four slices on four streams, across an auto-reset, against the fused step, bit for bit."""
    pc.check_pipeline_identical(make_sim, golden, n=512, steps=5, seed=13, all_reset_last=False, pipelines=(0, 1))


def test_mpr_option_pipelined_step_matches_fused(blobs, golden):
    """The -DSO101_MPR library (narrowphase="mpr", built on demand): the same sources with MPR's own portal depth instead of the EPA
    expansion - its launch chains and its fused step are bit-identical too, and it is a different library with a different build hash."""
    from so101_sim_amd import build
    assert build.source_hash(mpr=True) != build.source_hash() and build.lib_path(mpr=True) != build.lib_path()
    make = lambda n, seed=0, **cfg: ArraySim(blobs["f32"], n, backend="gpu", seed=seed, mpr=True, **cfg)
    pc.check_pipeline_identical(make, golden, n=8, steps=6, pipelines=(0, 1))


def test_single_env_with_more_candidates_than_pool_records(make_sim):
    pc.check_single_env_many_candidates(make_sim, n=1)
    pc.check_single_env_many_candidates(make_sim, n=3)


def test_more_contacts_than_the_solver_has_lanes_matches_the_oracle_under_the_same_rule(make_sim, blobs):
    """the 64-contact capacity is a deviation MuJoCo does not have; with the rule mirrored in the oracle
    (Oracle.set_contact_capacity) a state above the capacity is compared contact by contact, and the unlimited oracle shows what was given up."""
    from so101_sim_amd import native
    unlimited, kept, err = pc.check_contact_capacity_mirrored(make_sim, blobs, max_contacts=native.load_library().so101_max_contacts())
    assert unlimited > 64 >= kept and err <= 1e-3


def test_probe_outlier_states_of_round_6(make_sim, blobs, golden):
    pc.check_probe_outliers(make_sim, blobs, golden)


@pytest.mark.parametrize("switch", ["SO101_NO_SBT", "SO101_NO_HL"])
def test_support_tables_change_no_contact(blobs, switch):
    """Round 6.  SO101_NO_SBT: the broadphase drops a candidate pair when a hull's support-bound table (DevModel::hull_sbt) proves a separating
    plane beyond the oriented boxes' fifteen axes - a dropped pair would have ended in "no intersection".  SO101_NO_HL: k_narrow serves a flat
    face against a hull from the few vertices that can win in the cube-map cell of the face normal (DevModel::hl_entry) instead of the whole
    hull - the same support points.  Neither may change anything downstream: two handles, one created with the switch set (no tables), the same
    seed and random actions across a time limit - states and task outputs bit for bit (and fewer candidates with the support-bound tables)."""
    import os
    n, steps = 512, 30
    rng = np.random.RandomState(4)
    lo = np.array([-np.pi, -3.14158, -3.14158, -3.14158, -3.14158, 0.0], dtype=np.float32)
    hi = np.array([np.pi, 3.14158, 3.14158, 3.14158, 3.14158, 0.08], dtype=np.float32)
    acts = rng.uniform(lo, hi, size=(steps, n, 6)).astype(np.float32)
    traces, cands = [], []
    for off in (False, True):
        if off:
            os.environ[switch] = "1"
        try:
            sim = ArraySim(blobs["f32"], n, backend="gpu", seed=3, last_step=12, settle_max_substeps=200)
        finally:
            os.environ.pop(switch, None)
        sim.reset()
        tr, nc = [], 0.0
        for t in range(steps):
            obs, rew, disc, st = sim.step(acts[t])
            tr.append(np.concatenate([obs.ravel(), rew, disc, st.astype(np.float32)] + [a.ravel() for a in sim.get_state()]))
            nc += float(sim.get_diag()[:, 3].mean())
        traces.append(tr); cands.append(nc / steps)
    for t, (a, b) in enumerate(zip(*traces)):
        assert np.all(np.isfinite(a))
        np.testing.assert_array_equal(a, b, err_msg=f"step {t}")
    assert cands[0] < 0.95 * cands[1] if switch == "SO101_NO_SBT" else cands[0] == cands[1], cands


def test_three_launch_chains_match_fused(make_sim, golden):
    """n >= 64: so101_step cuts the cost-sorted envs into three slices on separate streams; still bit-identical to the
    fused single-launch step (different random actions per env, so the slices really differ in cost)."""
    pc.check_pipeline_identical(make_sim, golden, n=160, steps=5, seed=4, all_reset_last=False)


def test_chained_step_matches_fused_at_4096_envs(make_sim, golden):
    """The per-env chained step (one persistent launch, device-side queues, 2048 wavefronts on all eight XCDs) against the
    fused step, bit for bit, at the headline batch size: any stale hand-off between wavefronts shows up here."""
    pc.check_pipeline_identical(make_sim, golden, n=4096, steps=8, seed=11, all_reset_last=False, pipelines=(0, 2))


def test_merged_launches_match_fused(make_sim, golden):
    """pipeline = 3: the narrowphase of substep s + 1 rides in the solve launch of substep s (chunks through a per-chain
    device-side queue); four chains at 4096 envs, bit for bit against the fused step."""
    pc.check_pipeline_identical(make_sim, golden, n=4096, steps=8, seed=11, all_reset_last=False, pipelines=(0, 3))
    pc.check_pipeline_identical(make_sim, golden, n=8, steps=6, pipelines=(0, 3))


def test_chained_step_with_few_wavefronts(make_sim, golden):
    """Eight persistent wavefronts for 160 envs: every wavefront alternates between narrowphase chunks and solve items of many
    envs (queue wrap, the CAS paths, idle polling)."""
    pc.check_pipeline_identical(make_sim, golden, n=160, steps=5, seed=4, all_reset_last=False, pipelines=(0, 2), chain_waves=8)


# ---- SO100HandOverPen: same kernels, second scene blob (pen + utensil holder, two overlap boxes)
@pytest.fixture(scope="module")
def make_pen(blobs_pen):
    def f(n, seed=0, **cfg):
        return ArraySim(blobs_pen["f32"], n, backend="gpu", seed=seed, **cfg)
    return f


def test_pen_forward_stages(make_pen, blobs_pen):
    pc.check_forward_stages(make_pen, blobs_pen, n=8)


def test_pen_control_step(make_pen, blobs_pen):
    pc.check_control_step(make_pen, blobs_pen, n=8, iterations=100)


def test_pen_reward_bitexact(make_pen, blobs_pen):
    pc.check_reward_generic(make_pen, blobs_pen, n=256)


def test_pen_env_semantics(make_pen, blobs_pen):
    from oracle.oracle import Oracle
    o = Oracle(blobs_pen["f64"])
    o.env_config(seed=11, env_id=100)
    o.env_reset()
    pc.check_env_semantics(make_pen, blobs_pen, n=4, settle=200, steps=9, last_step=7, iterations=50, rest_z=float(o.get_state()[0][8]))


def test_full_settle_matches_oracle(make_sim, blobs):
    """reset with the reference's full 1000-substep settle budget: rest pose vs oracle, KAT-2 heights."""
    from oracle.oracle import Oracle
    sim = make_sim(3, seed=21)
    sim.reset()
    q, v, _ = sim.get_state()
    for e in range(3):
        o = Oracle(blobs["f64"])
        o.env_config(seed=21, env_id=e)
        o.env_reset()
        qo, vo, _ = o.get_state()
        assert np.abs(q[6:9, e] - qo[6:9]).max() < 5e-4 and np.abs(q[13:16, e] - qo[13:16]).max() < 5e-4
        assert abs(q[8, e] - qo[8]) < 2e-5                       # banana rest height
    assert np.all(q[:6] == 0) and np.all(v[:6] == 0)


def test_determinism_and_shard_invariance(make_sim, blobs):
    """Same seed => bit-identical trajectories; per-env RNG keyed by GLOBAL env id, so a shard starting at
    env_id_base=k reproduces envs k.. of the unsharded run."""
    n, steps = 8, 3
    rng = np.random.RandomState(0)
    acts = rng.uniform(-0.5, 0.5, size=(steps, n, 6)).astype(np.float32)

    def run(count, base, a):
        s = make_sim(count, seed=5, settle_max_substeps=100, solver_iterations=20, env_id_base=base)
        s.reset()
        for t in range(steps):
            s.step(a[t])
        return s.get_state()[0]
    full = run(n, 0, acts)
    again = run(n, 0, acts)
    np.testing.assert_array_equal(full, again)
    shard = run(n // 2, n // 2, acts[:, n // 2:])
    np.testing.assert_array_equal(full[:, n // 2:], shard)


def test_properties_at_benchmark_size(make_sim, blobs):
    """4096 envs (BASELINE.json configs[1]): invariants that do not need the oracle."""
    n = 4096
    sim = make_sim(n, seed=1, solver_iterations=20, last_step=500)
    sim.reset()
    q, v, _ = sim.get_state()
    assert np.all(np.isfinite(q)) and np.all(np.isfinite(v))
    assert np.all(q[:6] == 0)
    np.testing.assert_allclose(np.linalg.norm(q[9:13], axis=0), 1.0, atol=1e-5)
    np.testing.assert_allclose(np.linalg.norm(q[16:20], axis=0), 1.0, atol=1e-5)
    lo = np.array([-np.pi, -3.14158, -3.14158, -3.14158, -3.14158, 0.0], dtype=np.float32)
    hi = np.array([np.pi, 3.14158, 3.14158, 3.14158, 3.14158, 0.08], dtype=np.float32)
    rng = np.random.RandomState(2)
    total = np.zeros(n)
    prev_err = np.zeros(n, dtype=bool)
    for t in range(10):
        act = rng.uniform(lo, hi, size=(n, 6)).astype(np.float32)
        obs, rew, disc, st = sim.step(act)
        total += rew
        assert set(np.unique(rew)) <= {0.0, 1.0} and set(np.unique(st[~prev_err])) <= {1, 2} and np.all(st[prev_err] == 0)
        # discount 0 without reward = physics error (full-range random targets spin a prop up now and then: ~1e-4 per
        # env-step): rare, flagged, ends the episode
        err = (disc == 0) & (rew == 0)
        assert err.sum() <= 4 and np.all(st[err] == 2) and np.all(sim.get_diag()[err, 4] & 8)
        assert np.all(disc[(rew == 0) & ~err] == 1.0)
        np.testing.assert_array_equal(obs[~prev_err, 12:18], act[~prev_err])      # (a FIRST step ignores its action)
        prev_err = err
    q, v, _ = sim.get_state()
    assert np.all(np.isfinite(q)) and np.all(np.isfinite(v))
    rlo = np.array([-2.2, -3.14158, 0, -2, -3.14158, -0.2])[:, None]
    rhi = np.array([2.2, 0.2, 3.14158, 1.8, 3.14158, 2])[:, None]
    # joint limits are soft rows (solref 0.02: stiffness ~2.8e3 s^-2) against 35 N.m motors on ~0.1 kg m^2 links
    # arriving at up to ~90 rad/s: the overshoot is v / sqrt(k) ~ 1.7 rad at worst, bounded but not small
    assert np.all(q[:6] > rlo - 2.5) and np.all(q[:6] < rhi + 2.5)
    d = sim.get_diag()
    assert np.all(d[:, 4] & 7 == 0), "contact/candidate overflow at benchmark size"
    ep = sim._get(sim.ep_return)
    alive = sim._get(sim.step_count) == 10               # (an env that ended early has been reset: its return restarted)
    np.testing.assert_array_equal(ep[alive], total.astype(np.float32)[alive])


def test_divergence_handling(make_sim, blobs):
    pc.check_divergence_handling(make_sim, blobs)


def test_contact_rich_states(make_sim, blobs, golden):
    pc.check_contact_rich(make_sim, blobs, golden, count=24)


@pytest.mark.parametrize("task_name", ["SO100HandOverBanana", "SO100HandOverPen"])
def test_python_dropin_api(task_name):
    """task_suite.create_task_env keeps the reference's contract (task_suite.py:103-155) for both hand-over tasks:
    dm_env TimeSteps from the single env, tensors with a leading env dimension from the batched extension."""
    import torch
    from so101_sim_amd import task_suite
    env = task_suite.create_task_env(task_name, time_limit=0.2, random_state=7, settle_max_substeps=100)
    ts = env.reset()
    assert ts.first() and ts.reward is None and ts.discount is None
    assert set(ts.observation) >= {"joints_pos", "undelayed_joints_pos", "commanded_joints_pos", "physics_state"}
    spec = env.action_spec()
    assert spec.shape == (6,)
    n = 0
    while not ts.last():
        ts = env.step(np.zeros(6, dtype=np.float32))
        n += 1
    assert n == 10 and ts.discount == 1.0 and ts.reward == 0.0        # time limit 0.2 s = 10 control steps of 20 ms
    assert env.step(np.zeros(6, dtype=np.float32)).first()             # auto-reset
    benv = task_suite.create_task_env(task_name, time_limit=10.0, random_state=7, n_envs=16, settle_max_substeps=100)
    benv.reset_all()
    out = benv.step_tensor(torch.zeros(16, 6, device="cuda"))
    assert all(torch.isfinite(t.float()).all() for t in out if torch.is_tensor(t))


def test_lerobot_wrapper_on_gpu():
    """SO101LeRobotWrapper (so101_lerobot_wrapper.py:60-122) over the batched env: observation.state is the delayed
    joints_pos of the underlying env, the action is echoed, frame/timestamp count wrapper steps."""
    import torch
    from so101_sim_amd.lerobot import SO101LeRobotWrapper
    w = SO101LeRobotWrapper(time_limit=10.0, n_envs=8, random_state=3, settle_max_substeps=100)
    o = w.reset()
    assert o["observation.state"].shape == (8, 6) and torch.all(o["observation.state"] == 0) and torch.all(o["action"] == 0)
    act = torch.full((8, 6), 0.1)
    for t in range(1, 8):
        o = w.step(act)
        assert torch.all(o["frame_index"] == t) and torch.allclose(o["timestamp"], torch.full((8,), 0.1 * t, device=o["timestamp"].device))
        assert torch.equal(o["observation.state"], w.env.obs[:, 0:6]) and torch.allclose(o["action"].cpu(), act)
        if t <= 5:
            assert torch.all(o["observation.state"] == 0)              # 5-step observation delay
    assert torch.any(o["observation.state"] != 0)
    w1 = SO101LeRobotWrapper(time_limit=10.0, n_envs=1, random_state=3, settle_max_substeps=100)
    ep = w1.collect_episode([np.zeros(6, dtype=np.float32)] * 3)
    assert len(ep) == 4 and ep[0]["observation.state"].shape == (6,) and ep[3]["frame_index"].item() == 3 and w1.episode_index == 1
