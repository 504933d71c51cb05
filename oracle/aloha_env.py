"""ORACLE - TEST INFRASTRUCTURE ONLY: the env layer of the ALOHA hand-over tasks on top of the fp64 oracle's physics.

Restates, in plain Python, what dm_control's Environment does around physics.step() for `HandOver` (reference
so101_sim/tasks/hand_over.py:122-349 on so101_sim/tasks/base/aloha2_task.py:145-444):
    before_step     aloha2_task.py:316-349   ctrl = action, grippers converted follower -> sim_ctrl, no clipping
    observables     aloha2_task.py:386-444   joints_pos (six joints + the left finger in follower units, per arm), joints_vel (all 16
                                             joint velocities), commanded_joints_pos; delays of 0.1 s = 5 control steps
                                             (aloha2_task.py:104-105,223-243), buffers padded with the reset value (task_suite.py:154)
    reward          hand_over.py:246-284     overlap mode (the default): 0 while a prop moves, 1 when the object's box overlaps
                                             every overlap box of the container
    termination     aloha2_task.py:355-367   reward >= 1 ends the episode with discount 0; the time limit with discount 1
    reset           aloha2_task.py:369-383, hand_over.py:340-346   (placement + settle: so101_oracle.cpp env_reset)
Nothing under so101_sim_amd/ imports this file."""
from __future__ import annotations

import collections

import numpy as np

from oracle.oracle import Oracle

GRIP = dict(sim_qpos=(0.037, 0.0078), sim_ctrl=(0.037, 0.002), follower=(1.5155, -0.06135))      # (open, close), aloha2_task.py:57-70
OBS_QPOS = [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14]


def convert_gripper(v, a, b):
    (ao, ac), (bo, bc) = GRIP[a], GRIP[b]
    return (v - ac) / (ao - ac) * (bo - bc) + bc


class AlohaOracleEnv:
    def __init__(self, blob_f64: bytes, seed=0, env_id=0, last_step=1 << 30, settle_max_substeps=1000, reward_based_on_overlap=True,
                 reward_requires_handover=False, geom_class=None, bodies=None, dist_threshold=0.0, n_substeps=10, reward_touching=False,
                 prop_dofadr=(16, 22)):
        """geom_class / bodies = (object body, container body) / dist_threshold: the model's task_geom_class, task bodies and
        task_dist_threshold, needed by the contact-sequence reward only.  reward_touching: the 'contact' reward of the Dining tasks
        (dining_place_in_container.py:126-154) - 1 when a geom of the object (class 1) touches a geom of the receptacle (class 2) and
        neither prop moves; prop_dofadr = first dof of (object, receptacle)."""
        self.o = Oracle(blob_f64)
        self.overlap, self.requires_handover = reward_based_on_overlap, reward_requires_handover
        self.geom_class, self.bodies, self.dist_threshold = geom_class, bodies, dist_threshold
        self.success_state = 2
        self.touching, self.prop_dofadr = reward_touching, prop_dofadr
        self.n_substeps = n_substeps      # physics steps per control step (10 in the reference; the emulated CPU tests shorten it)
        self.o.env_config(seed=seed, env_id=env_id, settle_max_substeps=settle_max_substeps)
        self.last_step = last_step
        self.need_reset = True
        self.step_count = 0

    # -- observation pieces
    def _pos(self):
        q = self.o.get_state()[0]
        p = q[OBS_QPOS].copy()
        p[6], p[13] = convert_gripper(q[6], "sim_qpos", "follower"), convert_gripper(q[14], "sim_qpos", "follower")
        return p

    def _vel(self):
        return self.o.get_state()[1][:16].copy()

    def _cmd(self):
        c = self.ctrl.copy()
        c[6], c[13] = convert_gripper(c[6], "sim_ctrl", "follower"), convert_gripper(c[13], "sim_ctrl", "follower")
        return c

    def _obs(self, delayed_pos, delayed_vel):
        return np.concatenate([delayed_pos, delayed_vel, self._pos(), self._vel(), self._cmd()])

    def begin(self, qpos, qvel, warm, ctrl):
        """adopt a post-reset state (for side-by-side runs against another implementation's reset)"""
        self.o.set_state(qpos, qvel, warm)
        self.ctrl = np.array(ctrl, dtype=np.float64)
        self.o.set_ctrl(self.ctrl)
        self.ring_pos = collections.deque([self._pos()] * 5, maxlen=5)
        self.ring_vel = collections.deque([self._vel()] * 5, maxlen=5)
        self.step_count, self.need_reset = 0, False
        self.success_state = 0 if self.requires_handover else 2

    def reset(self):
        self.o.env_reset()
        q, v, w = self.o.get_state()
        self.begin(q, v, w, np.concatenate([[0.0, -0.96, 1.16, 0.0, -0.3, 0.0, 0.002]] * 2))
        return self._obs(self.ring_pos[0], self.ring_vel[0])

    def _contact_reward(self):
        """hand_over.py:286-338: the three-state sequence over physics.data.contact (the contacts of the state after physics.step():
        the caller has run forward() there); every call advances the state machine"""
        cls = self.geom_class
        pairs = [(int(cls[c["geom1"]]), int(cls[c["geom2"]])) for c in self.o.contacts()]
        touching = lambda a, b: any(((c1 & a) and (c2 & b)) or ((c2 & a) and (c1 & b)) for c1, c2 in pairs)
        q, v, _ = self.o.get_state()
        moving = max(np.abs(v[16:19]).max(), np.abs(v[22:25]).max()) >= 1e-3
        if self.success_state == 0:
            if touching(8, 1):
                self.success_state = 1
        elif self.success_state == 1:
            if touching(4, 1):
                self.success_state = 2
        else:
            po, pc = self.o.body_pose(self.bodies[0])[0], self.o.body_pose(self.bodies[1])[0]
            inside = np.hypot(pc[0] - po[0], pc[1] - po[1]) < self.dist_threshold
            if not moving and touching(1, 2) and inside:
                return 1.0
        return 0.0

    def _touch_reward(self):
        """dm_control reads physics.data.contact after physics.step(), whose legacy step ends with mj_step1: the contacts of the state
        AFTER the last substep (forward() recomputes them there)."""
        v = self.o.get_state()[1]
        a, b = self.prop_dofadr
        if max(np.abs(v[a:a + 3]).max(), np.abs(v[b:b + 3]).max()) >= 1e-3:        # any_props_moving: linear velocities only
            return 0.0
        self.o.forward()
        cls = self.geom_class
        for c in self.o.contacts():
            c1, c2 = int(cls[c["geom1"]]), int(cls[c["geom2"]])
            if ((c1 & 1) and (c2 & 2)) or ((c2 & 1) and (c1 & 2)):
                return 1.0
        return 0.0

    def step(self, action):
        """-> obs, reward, discount, step_type"""
        if self.need_reset:
            return self.reset(), 0.0, 1.0, 0
        a = np.asarray(action, dtype=np.float64)
        self.ctrl = a.copy()
        self.ctrl[6], self.ctrl[13] = convert_gripper(a[6], "follower", "sim_ctrl"), convert_gripper(a[13], "follower", "sim_ctrl")
        self.o.set_ctrl(self.ctrl)
        diverged = self.o.substeps(self.n_substeps, False)
        self.step_count += 1
        dp, dv = self.ring_pos[0], self.ring_vel[0]          # the value of control step k - 5
        self.ring_pos.append(self._pos()); self.ring_vel.append(self._vel())
        # dm_control's Environment.step: reward = get_reward(); discount = get_discount() -> should_terminate_episode() -> get_reward();
        # terminating = should_terminate_episode() -> get_reward()  (aloha2_task.py:353-367): three evaluations, and the contact
        # sequence advances its state on each of them
        if diverged:
            r = r_disc = r_term = 0.0
        elif self.touching:
            r = r_disc = r_term = self._touch_reward()
        elif self.overlap:
            r = r_disc = r_term = float(self.o.reward())
        else:
            self.o.forward()
            r = self._contact_reward(); r_disc = self._contact_reward(); r_term = self._contact_reward()
        disc0, success, timeout = r_disc >= 1.0 or bool(diverged), r_term >= 1.0 or bool(diverged), self.step_count >= self.last_step
        st = 2 if (success or timeout) else 1
        self.need_reset = st == 2
        return self._obs(dp, dv), r, 0.0 if disc0 else 1.0, st
