"""Stage clocks of k_pipe_solve (so101_debug_stages) on the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from so101_sim_amd.model import scenes
from tests.simharness import ArraySim
raw32, meta = scenes.load_blob("banana", "f32")
N = int(os.environ.get("STAGES_N", "4096"))
s = ArraySim(raw32, N, backend="gpu", seed=0, settle_max_substeps=300, last_step=100000, prefetch_resets=0)
s.reset()
lo = np.array([-np.pi, -3.14158, -3.14158, -3.14158, -3.14158, 0.0], dtype=np.float32)
hi = np.array([np.pi, 3.14158, 3.14158, 3.14158, 3.14158, 0.08], dtype=np.float32)
rng = np.random.RandomState(2)
for t in range(30):
    s.step(rng.uniform(lo, hi, size=(N, 6)).astype(np.float32))
st = torch.zeros(N, 8, dtype=torch.int32, device=s.dev)
s.sim.debug_stages(st.data_ptr()); torch.cuda.synchronize()
st = st.cpu().numpy().astype(np.int64)
names = ["smooth", "gather", "constraints", "solver", "integrate", "next broadphase"]
us = st[:, :6] * 1e-2
print("per-env stage time of one substep (us): mean / p50 / p99 / max")
for i, n in enumerate(names):
    print("  %-16s %7.1f %7.1f %7.1f %7.1f" % (n, us[:, i].mean(), np.median(us[:, i]), np.percentile(us[:, i], 99), us[:, i].max()))
tot = us.sum(1)
print("  %-16s %7.1f %7.1f %7.1f %7.1f" % ("total", tot.mean(), np.median(tot), np.percentile(tot, 99), tot.max()))
print("sum over envs / 1792 resident waves = %.3f ms ; max env %.3f ms" % (tot.sum() * 1e-3 / 1792, tot.max() * 1e-3))
ncon, it = st[:, 6], st[:, 7]
for lo_, hi_ in ((0, 1), (1, 5), (5, 9), (9, 13), (13, 20), (20, 33)):
    mk = (ncon >= lo_) & (ncon < hi_)
    if mk.any():
        print("  ncon [%2d,%2d): %4d envs  constraints %.1f  solver %.1f us  iters %.1f" % (lo_, hi_, mk.sum(), us[mk, 2].mean(), us[mk, 3].mean(), it[mk].mean()))

order = np.argsort(-tot)
print("slowest envs: total us | smooth gather constraints solver integrate next | ncon iters")
for e in order[:16]:
    print("  %6.1f | %s | %3d %3d" % (tot[e], " ".join("%6.1f" % x for x in us[e]), ncon[e], it[e]))
print("iterations histogram:", np.bincount(it.astype(int))[:25])
print("solver us per iteration by ncon bucket:")
for lo_, hi_ in ((1, 5), (5, 9), (9, 13), (13, 20), (20, 33), (33, 65)):
    mk = (ncon >= lo_) & (ncon < hi_) & (it > 0)
    if mk.any():
        print("  ncon [%2d,%2d): %.1f us/iter" % (lo_, hi_, (us[mk, 3] / it[mk]).mean()))

nc_ = torch.zeros(N, dtype=torch.int32, device=s.dev); cand_ = torch.zeros(N, 256, dtype=torch.int32, device=s.dev)
tk_ = torch.zeros(N, 256, dtype=torch.int32, device=s.dev); cr_ = torch.zeros(N, 256, 24, device=s.dev)
s.sim.debug_candidates(nc_.data_ptr(), cand_.data_ptr(), tk_.data_ptr(), cr_.data_ptr()); torch.cuda.synchronize()
ph = tk_.cpu().numpy()[:, 240:256].astype(np.int64) * 1e-2
print("solver phases (us per env-substep, mean; per iteration): setup gradient hessian factor solve linesearch cost")
print("   mean  ", " ".join("%6.2f" % x for x in ph[:, :7].mean(0)), " total %.1f" % ph[:, :7].sum(1).mean())
print("   /iter ", " ".join("%6.2f" % x for x in (ph[:, :7] / np.maximum(it, 1)[:, None]).mean(0)))
print("next-substep phase (us, mean): state store %.2f  kinematics %.2f  geom boxes %.2f  pair list %.2f  oriented boxes %.2f  publish %.2f" % tuple(ph[:, 8:14].mean(0)))
print("pair list split (us, mean): burst load of the pair words %.2f  box tests %.2f  (compaction = pair list - both)" % tuple(ph[:, 14:16].mean(0)))
