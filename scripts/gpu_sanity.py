"""GPU sanity: HIP kernels vs the CPU oracle on a handful of states + first timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from so101_sim_amd.model import scenes
from oracle.oracle import Oracle
from tests.simharness import ArraySim

raw64, meta = scenes.load_blob('banana', 'f64'); raw32, _ = scenes.load_blob('banana', 'f32')
kat_q = np.array([2.65129794e-01, 4.09270959e-03, 4.21711098e-01, 9.99246834e-01,-1.69673020e-04, 9.86495446e-05, 3.88036861e-02,
 -2.17988678e-01,-3.96717335e-02, 4.22621829e-01, 9.99999456e-01,-5.82714664e-04,-8.65621822e-04,-8.25292623e-06])
N = 4
rng = np.random.RandomState(0)
Q = np.zeros((20, N)); V = np.zeros((18, N)); CT = np.zeros((6, N))
for e in range(N):
    Q[:6, e] = rng.uniform(-0.5, 0.5, 6) if e else 0
    Q[6:, e] = kat_q
    V[:6, e] = rng.uniform(-1, 1, 6) if e else 0
    CT[:, e] = rng.uniform(-1, 1, 6) if e else [28, 42, 18, -21, 1009, -157.5]
s = ArraySim(raw32, N, backend='gpu')
s.set_state(Q, V, CT, np.zeros((18, N)))
d = s.debug_forward()
for e in range(N):
    o = Oracle(raw64); o.set_state(Q[:, e], V[:, e], np.zeros(18)); o.set_ctrl(CT[:, e]); o.forward()
    a, asm = o.qacc()
    print(e, 'M', np.abs(d[e]['M'] - o.M()[:6, :6]).max(), 'smooth rel', np.abs(d[e]['qacc_smooth'] - asm).max() / np.abs(asm).max(),
          'ncon', d[e]['ncon'], len(o.contacts()), 'iters', d[e]['iters'], o.solver_iter, 'qacc rel', np.abs(d[e]['qacc'] - a).max() / np.abs(a).max(), 'ovf', d[e]['overflow'])
# one control step
s.physics(10)
q1, v1, _ = s.get_state()
for e in range(N):
    o = Oracle(raw64); o.set_state(Q[:, e], V[:, e], np.zeros(18)); o.set_ctrl(CT[:, e]); o.substeps(10)
    qo, vo, _ = o.get_state()
    print(e, 'step dq', np.abs(q1[:, e] - qo).max(), 'dv', np.abs(v1[:, e] - vo).max())
exp_q = np.array([5.85192160e-02, 5.80983147e-02, 6.58658498e-02,-8.00624348e-02, 7.67682376e-02,-7.65953670e-02])
print('KAT-1 fp32 rel err', np.abs((q1[:6, 0] - exp_q) / exp_q).max())
# timing at N=4096
for N2, iters in ((4096, 100), (4096, 10)):
    s2 = ArraySim(raw32, N2, backend='gpu', solver_iterations=iters)
    Q2 = np.tile(Q[:, :1], (1, N2)); s2.set_state(Q2, np.zeros((18, N2)), np.tile(CT[:, 1:2], (1, N2)), np.zeros((18, N2)))
    s2.physics(10); torch.cuda.synchronize()
    t = time.time(); 
    for _ in range(5): s2.physics(10)
    torch.cuda.synchronize(); dt = (time.time() - t) / 5
    print(f'N={N2} iters={iters}: {dt*1e3:.2f} ms per control step -> {N2/dt:.0f} env-steps/s', s2.get_diag()[:2])
# reset timing
s3 = ArraySim(raw32, 256, backend='gpu', solver_iterations=20)
t = time.time(); s3.reset(); torch.cuda.synchronize(); print('reset 256 envs', time.time() - t)
q3, v3, _ = s3.get_state(); print(q3[6:, :3].T, np.abs(v3[6:]).max(), s3.get_diag()[:3])
