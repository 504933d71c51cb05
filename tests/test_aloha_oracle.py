"""ALOHA - the general-tree model compile and the fp64 oracle on it, against the reference's OWN numeric tests
for this robot (so101_sim/tasks/test/aloha2_task_test.py): the first facts about contact dynamics that come from reference
test code rather than from notebooks.  CPU only: this file pins the ORACLE; the ALOHA kernels (the general-tree engine,
DESIGN.md section 8) are checked against that oracle in tests/test_tree_parity.py.

    aloha2_task_test.py:55-72   the home pose lies inside the action spec
    aloha2_task_test.py:75-101  close the gripper for 100 control steps: ctrl[6] == 0.002, qpos[6] == 0.0078 +- 0.001
                                (the finger meshes stop each other), joints_pos[6] == FOLLOWER_GRIPPER_CLOSE +- 0.01
    aloha2_task_test.py:103-114 open it for 50 steps: qpos[6] >= 0.035; joints_pos[6] in [1.55, 1.62]  <- NOT reproduced: 1.478
                                under EPA (the default), 1.475 under MPR; see test_open_gripper (strict xfail beside it)
"""
import numpy as np
import pytest

from oracle.oracle import Oracle
from so101_sim_amd.model import blob as blobfmt
from so101_sim_amd.model import scenes

# tasks/base/aloha2_task.py:115-132
ALL_JOINTS = [f"{side}/{j}" for side in ("left", "right")
              for j in ("waist", "shoulder", "elbow", "forearm_roll", "wrist_angle", "wrist_rotate", "left_finger", "right_finger")]
LIM = scenes.ALOHA_GRIPPER_LIMITS


def convert_gripper(v, a, b):          # aloha2_task.py:304-314
    (ao, ac), (bo, bc) = LIM[a], LIM[b]
    return (v - ac) / (ao - ac) * (bo - bc) + bc


def before_step(action):               # aloha2_task.py:316-349: joints as they are, grippers follower -> sim_ctrl
    ctrl = np.array(action, dtype=np.float64)
    ctrl[6] = convert_gripper(action[6], "follower", "sim_ctrl")
    ctrl[13] = convert_gripper(action[13], "follower", "sim_ctrl")
    return ctrl


@pytest.fixture(scope="module")
def bare():
    raw, meta = scenes.load_aloha_blob(None)
    return raw, meta, blobfmt.unpack(raw)


def _reset(o):                         # aloha2_task.py:369-380
    o.set_state(np.concatenate([scenes.ALOHA_HOME_QPOS, scenes.ALOHA_HOME_QPOS]), np.zeros(16), np.zeros(16))
    o.set_ctrl(np.concatenate([scenes.ALOHA_HOME_CTRL, scenes.ALOHA_HOME_CTRL]))


def test_compile_constants(bare):
    raw, meta, m = bare
    assert (int(m["nq"][0]), int(m["nv"][0]), int(m["nu"][0]), int(m["neq"][0])) == (16, 16, 14, 2)
    assert meta["joint_names"] == ALL_JOINTS                       # qpos order of aloha2_task.py:115-132
    assert list(m["jnt_type"]) == [1, 1, 1, 1, 1, 1, 3, 3] * 2       # six hinges and two slide fingers per arm
    np.testing.assert_allclose(m["act_gain"], [43, 265, 227, 78, 37, 10.4, 2000] * 2)              # aloha_pbr.xml:47-100
    np.testing.assert_allclose(np.asarray(m["act_bias"]).reshape(14, 3)[6], [0, -2000, -124])
    np.testing.assert_allclose(np.asarray(m["act_ctrlrange"]).reshape(14, 2)[6], [0.002, 0.037])
    np.testing.assert_allclose(np.asarray(m["jnt_actfrcrange"]).reshape(16, 2)[1], [-144, 144])
    np.testing.assert_allclose(m["dof_damping"][:8], [5.76, 20.0, 18.49, 6.78, 6.28, 1.2, 40, 40])
    np.testing.assert_allclose(m["dof_armature"][:8], [0, 0.395, 0.383, 0.14, 0.008, 0, 0.243, 0.243])
    np.testing.assert_allclose(m["dof_frictionloss"][:8], [0, 2.0, 1.15, 0, 0, 0, 0, 0])
    assert list(np.asarray(m["eq_dof"])) == [6, 7, 14, 15]
    act_dof = list(m["act_dof"])
    assert act_dof == [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14]      # the gripper actuator drives the LEFT finger of each arm
    # the table top: body z -0.732 + 0.011 (aloha2_task.py:107,493-496), box centre +0.6509, half height 0.1 (scene_pbr.xml:70)
    names = meta["geom_names"]
    t = names.index("table")
    body = int(np.asarray(m["geom_body"])[t])
    top = np.asarray(m["body_pos"]).reshape(-1, 3)[body][2] + np.asarray(m["geom_pos"]).reshape(-1, 3)[t][2] + np.asarray(m["geom_size"]).reshape(-1, 3)[t][2]
    assert abs(top - 0.0299) < 1e-9
    key = meta["keyframes"]["neutral_pose"]
    np.testing.assert_allclose(key["qpos"][:8], [0, -0.96, 1.16, 0, -0.3, 0, 0.0084, 0.0084])


def test_home_pose_is_inside_the_action_spec(bare):
    _, _, m = bare
    cr = np.asarray(m["act_ctrlrange"]).reshape(14, 2)
    lo, hi = cr[:, 0].copy(), cr[:, 1].copy()
    lo[[0, 7]], hi[[0, 7]] = -np.pi / 2, np.pi / 2                 # aloha2_task.py:290-291
    lo[[6, 13]], hi[[6, 13]] = LIM["follower"][1], LIM["follower"][0]
    for off in (0, 7):
        assert np.all(lo[off:off + 6] < scenes.ALOHA_HOME_CTRL[:6]) and np.all(hi[off:off + 6] > scenes.ALOHA_HOME_CTRL[:6])
        assert np.all(lo[off:off + 6] < scenes.ALOHA_HOME_QPOS[:6]) and np.all(hi[off:off + 6] > scenes.ALOHA_HOME_QPOS[:6])


def test_close_gripper(bare):
    raw, _, _ = bare
    o = Oracle(raw)
    _reset(o)
    action = np.concatenate([scenes.ALOHA_HOME_CTRL, scenes.ALOHA_HOME_CTRL])
    action[6] = LIM["follower"][1]
    action[0] = action[7] = 0.1
    ctrl = before_step(action)
    assert ctrl[6] == 0.002                                        # SIM_GRIPPER_CTRL_CLOSE
    o.set_ctrl(ctrl)
    for _ in range(100):
        o.substeps(10, False)
        q = o.get_state()[0]
        assert abs(q[6] - q[7]) < 1.5e-3                           # the joint equality keeps the two fingers together
    assert abs(q[6] - 0.0078) <= 0.001, q[6]                       # SIM_GRIPPER_QPOS_CLOSE: the finger meshes stop each other
    assert abs(convert_gripper(q[6], "sim_qpos", "follower") - LIM["follower"][1]) <= 0.01
    assert abs(q[0] - 0.1) < 1e-3 and abs(q[8] - 0.1) < 1e-3       # the waists followed their targets


def _open_gripper_obs(raw, epa=True, collide=True):
    o = Oracle(raw)
    o.set_narrowphase(epa)
    o.set_collision(collide)
    _reset(o)
    action = np.zeros(14)
    action[6] = LIM["follower"][0]
    o.set_ctrl(before_step(action))
    for _ in range(50):
        o.substeps(10, False)
    q = o.get_state()[0]
    return q, convert_gripper(q[6], "sim_qpos", "follower"), o


def test_open_gripper(bare):
    """aloha2_task_test.py:103-114.  With zero targets both arms stretch out and their grippers JAM against each other in the
    middle of the table (reach 0.55 m each, bases 0.94 m apart; the right gripper, commanded 0 = almost closed, shuts on itself).
    The reference asserts qpos[6] >= 0.035 - reproduced - and joints_pos[6] in [1.55, 1.62], i.e. a finger 0.6-1.9 mm BEYOND its
    position target of 0.037 (kp 2000: 1.2-3.8 N pushing it outwards) - NOT reproduced, with either narrowphase (round 4: the test
    now runs the oracle's default, EPA = the minimum translation mujoco >= 3.3 reports, and the MPR option beside it):

        no collisions      qpos[6] 0.03700 -> joints_pos[6] 1.5153   (the target, exactly)
        EPA (default)      qpos[6] 0.03631 -> 1.478
        MPR option         qpos[6] 0.03626 -> 1.475

    The jam after 50 control steps under EPA (pair, depth, normal from the first geom into the second, position):
        left gripper_base prop bar  x right gripper_base prop bar   2.21 mm  ( 1.00  0.00  0.02)  (-0.000 -0.030 0.497)
        left prop bar               x right d405 camera body        0.30 mm  ( 0.98  0.00 -0.21)  ( 0.001 -0.032 0.496)
        left prop bar               x right left_finger             0.11 mm  ( 0.74  0.00 -0.68)  (-0.034 -0.004 0.379)
        left prop bar               x right right_finger            0.11 mm  ( 0.74  0.00 -0.68)  (-0.034 -0.034 0.379)
        left d405                   x right prop bar                0.29 mm  ( 0.98  0.00  0.21)  (-0.001 -0.006 0.496)
        left left_finger            x right prop bar                0.10 mm  ( 0.74  0.00  0.68)  ( 0.034 -0.062 0.379)
        left right_finger           x right prop bar                0.12 mm  ( 0.74  0.00  0.68)  ( 0.034  0.024 0.379)
        right gripper: two fingertip sphere pairs (0.6 mm spheres)  0.28 mm  ( 0.00 -1.00  0.00)  the closed right gripper
    Every normal on a LEFT finger lies in the x-z plane; the fingers slide along y.  The contacts therefore put no normal force
    along the finger joint - only friction (condim 4, mu 1) against its motion: the left finger creeps towards its target
    (0.0354 at step 15, 0.0363 at step 50, wrist_angle pushed to 0.46 rad) and stops 0.7 mm short, where kp (0.037 - q) = 1.4 N is
    held by friction on the two finger contacts.  For the reference's value something must push the finger 1.2-3.8 N OUTWARDS at
    t = 1 s; nothing in this contact set can (gravity along the joint: 5e-4 N).  Candidates that cannot be decided without
    MuJoCo: the arms still rebounding from the first impact at t = 1 s (an acceleration of 4-11 m/s^2 along y of the 0.33 kg
    finger + armature would do), or a different jam geometry.  The deviation is recorded here and asserted as a strict xfail below;
    what IS asserted is the oracle's own value, so that a change of the contact model shows up.

    Round 5: hull pairs resting on flat features now carry up to five contacts (hull_patch; the reference runs with multiccd,
    aloha2_task.py:197).  The jam holds the finger less firmly with the extra contacts sharing the load - EPA 1.478 -> 1.508 (the MPR option keeps
    one contact per hull pair and stays at 1.475), against 1.515 without collisions - but nothing pushes it BEYOND its target either: the reference's [1.55, 1.62] stays
    unreproduced (strict xfail below), now 0.04 away instead of 0.07."""
    raw, _, _ = bare
    q, obs, _ = _open_gripper_obs(raw)
    assert q[6] >= 0.035, q[6]                                     # aloha2_task_test.py:112
    assert 1.49 <= obs <= 1.53, obs                                # this oracle (EPA, hull patches: 1.5075); the reference: [1.55, 1.62], see test below
    q_mpr, obs_mpr, _ = _open_gripper_obs(raw, epa=False)
    assert q_mpr[6] >= 0.035 and 1.46 <= obs_mpr <= 1.50, obs_mpr          # (the MPR option keeps one contact per hull pair: 1.475)
    o1 = Oracle(raw); o1.set_hull_multicontact(False)                  # the single EPA contact per hull pair of rounds 1-4: 1.478
    _reset(o1)
    action = np.zeros(14); action[6] = LIM["follower"][0]
    o1.set_ctrl(before_step(action))
    for _ in range(50):
        o1.substeps(10, False)
    assert 1.46 <= convert_gripper(o1.get_state()[0][6], "sim_qpos", "follower") <= 1.50
    q_free, obs_free, _ = _open_gripper_obs(raw, collide=False)
    assert abs(q_free[6] - 0.037) < 1e-5 and abs(obs_free - 1.5153) < 1e-3


@pytest.mark.xfail(strict=True, reason="aloha2_task_test.py:113-114 (joints_pos[6] in [1.55, 1.62] with the grippers jammed) is NOT reproduced: "
                                       "1.508 under EPA with hull patches (1.478 with one contact per hull pair), 1.475 under MPR, 1.515 without collisions - see test_open_gripper")
def test_open_gripper_reference_bound(bare):
    raw, _, _ = bare
    _, obs, _ = _open_gripper_obs(raw)
    assert 1.55 <= obs <= 1.62, obs


def test_euler_step_with_joint_damping_is_implicit(bare):
    """mj_Euler with damping: qvel += h (M + h D)^-1 M qacc (numpy, from the oracle's own M and qacc)."""
    raw, _, m = bare
    o = Oracle(raw)
    rng = np.random.RandomState(0)
    q = np.concatenate([scenes.ALOHA_HOME_QPOS, scenes.ALOHA_HOME_QPOS]) + 0.05 * rng.normal(size=16) * np.array([1] * 6 + [0.02] * 2 + [1] * 6 + [0.02] * 2)
    v = rng.normal(size=16) * np.array([1] * 6 + [0.01] * 2 + [1] * 6 + [0.01] * 2)
    o.set_state(q, v, np.zeros(16))
    o.set_ctrl(np.concatenate([scenes.ALOHA_HOME_CTRL, scenes.ALOHA_HOME_CTRL]))
    o.forward()
    M, (qacc, _) = o.M(), o.qacc()
    D = np.asarray(m["dof_damping"])
    expect = v + 0.002 * np.linalg.solve(M + 0.002 * np.diag(D), M @ qacc)
    o.set_state(q, v, np.zeros(16))
    o.substeps(1, False)
    q1, v1, _ = o.get_state()
    np.testing.assert_allclose(v1, expect, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(q1, q + 0.002 * v1, rtol=0, atol=1e-14)
    assert np.abs(v1 - (v + 0.002 * qacc)).max() > 1e-4            # (and the damping matters at these rates)


def test_free_space_accelerations_against_numpy_dynamics(bare):
    """qacc_smooth of the two 8-dof trees against an independent numpy evaluation: M from the oracle (CRBA, checked against
    the compiler's own M(qpos0) below), bias by finite differences of the kinetic and potential energy - which covers the slide
    joints' Coriolis terms -, actuator and passive forces from the model numbers."""
    raw, meta, m = bare
    o = Oracle(raw)
    o.set_collision(False)
    o.set_state(np.zeros(16), np.zeros(16), np.zeros(16))
    o.forward()
    np.testing.assert_allclose(np.diag(o.M()), meta["M0_diag"], rtol=1e-12)
    rng = np.random.RandomState(1)
    q = np.concatenate([scenes.ALOHA_HOME_QPOS, scenes.ALOHA_HOME_QPOS]) + 0.2 * rng.normal(size=16) * np.array([1] * 6 + [0.02] * 2 + [1] * 6 + [0.02] * 2)
    v = rng.normal(size=16) * np.array([1] * 6 + [0.05] * 2 + [1] * 6 + [0.05] * 2)
    ctrl = np.concatenate([scenes.ALOHA_HOME_CTRL, scenes.ALOHA_HOME_CTRL])

    def mass(qq):
        o.set_state(qq, np.zeros(16), np.zeros(16)); o.forward(); return o.M() - np.diag(np.asarray(m["dof_armature"]))

    def potential(qq):
        o.set_state(qq, np.zeros(16), np.zeros(16)); o.forward()
        e = 0.0
        bm = np.asarray(m["body_mass"]); ip = np.asarray(m["body_ipos"]).reshape(-1, 3)
        from oracle import geomcheck as gc
        for b in range(1, len(bm)):
            if bm[b] > 0:
                p, qu = o.body_pose(b)
                e += bm[b] * 9.81 * (np.asarray(p) + gc.quat2mat(qu) @ ip[b])[2]
        return e
    # Lagrange: bias_i = sum_jk (dM_ij/dq_k - 0.5 dM_jk/dq_i) v_j v_k + dV/dq_i
    eps = 1e-6
    dM = np.zeros((16, 16, 16)); dV = np.zeros(16)
    for k in range(16):
        e = np.zeros(16); e[k] = eps
        dM[:, :, k] = (mass(q + e) - mass(q - e)) / (2 * eps)
        dV[k] = (potential(q + e) - potential(q - e)) / (2 * eps)
    bias = np.einsum("ijk,j,k->i", dM, v, v) - 0.5 * np.einsum("jki,j,k->i", dM, v, v) + dV
    o.set_state(q, v, np.zeros(16)); o.set_ctrl(ctrl); o.forward()
    np.testing.assert_allclose(o.bias(), bias, rtol=2e-5, atol=2e-6)
    gain = np.asarray(m["act_gain"]); ab = np.asarray(m["act_bias"]).reshape(14, 3); dof = np.asarray(m["act_dof"])
    cr = np.asarray(m["act_ctrlrange"]).reshape(14, 2)
    frc = np.zeros(16)
    frc[dof] = gain * np.clip(ctrl, cr[:, 0], cr[:, 1]) + ab[:, 1] * q[dof] + ab[:, 2] * v[dof]
    fr = np.asarray(m["jnt_actfrcrange"]).reshape(16, 2)
    frc = np.clip(frc, fr[:, 0], fr[:, 1])
    rhs = frc - np.asarray(m["dof_damping"]) * v - o.bias()
    np.testing.assert_allclose(o.qacc()[1], np.linalg.solve(o.M(), rhs), rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("name", ["banana", "pen"])
def test_hand_over_scene_resets_and_rests(name, golden):
    """HandOverBanana / HandOverPen (tasks/hand_over.py): nq 30, nv 28, nu 14; reset = arms at HOME_QPOS, object
    and container dropped from z 0.1 inside their boxes (hand_over.py:36-56), settled on the table top at z 0.0299.  The banana
    rests with the same 1.71 mm between its origin and the table top as in the SO100 scene's notebook pose (KAT-1)."""
    raw, meta = scenes.load_aloha_blob(name)
    m = blobfmt.unpack(raw)
    assert (int(m["nq"][0]), int(m["nv"][0]), int(m["nu"][0])) == (30, 28, 14)
    o = Oracle(raw)
    o.env_config(seed=3, env_id=0)
    o.env_reset()
    q, v, _ = o.get_state()
    np.testing.assert_allclose(q[:16], np.concatenate([scenes.ALOHA_HOME_QPOS, scenes.ALOHA_HOME_QPOS]), atol=1e-12)
    assert 0.12 - 5e-3 <= q[16] <= 0.18 + 5e-3 and -0.1 - 5e-3 <= q[17] <= 0.1 + 5e-3
    assert -0.18 - 5e-3 <= q[23] <= -0.12 + 5e-3 and -0.1 - 5e-3 <= q[24] <= 0.1 + 5e-3
    yaw = 2 * np.arctan2(q[22], q[19])
    assert -0.6 * np.pi - 2e-2 <= yaw <= -0.4 * np.pi + 2e-2
    assert np.abs(v[16:]).max() < 2e-2
    if name == "banana":
        rest = golden["kat1"]["observation"]["physics_state"][8] - 0.42       # SO100 scene: banana origin above its table top
        assert abs(q[18] - 0.0299 - rest) < 5e-5, (q[18] - 0.0299, rest)
    before = q.copy()
    for _ in range(10):
        o.substeps(10, False)                                       # arms hold the home pose, props stay put
    q, v, _ = o.get_state()
    assert np.abs(q[16:19] - before[16:19]).max() < 1e-3 and np.abs(q[23:26] - before[23:26]).max() < 1e-3
    assert np.abs(q[:6] - scenes.ALOHA_HOME_CTRL[:6]).max() < 0.03  # position servos against gravity: a few hundredths of a radian
    assert o.reward() == 0.0
