"""Per-candidate narrowphase time (k_narrow's ticks buffer) on the bench workload: distribution and the slowest pairs."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from so101_sim_amd.model import scenes
from tests.simharness import ArraySim
raw32, meta = scenes.load_blob("banana", "f32")
gn = meta["geom_names"]
N = 4096
s = ArraySim(raw32, N, backend="gpu", seed=0, settle_max_substeps=300, last_step=100000, prefetch_resets=0)
s.reset()
lo = np.array([-np.pi, -3.14158, -3.14158, -3.14158, -3.14158, 0.0], dtype=np.float32)
hi = np.array([np.pi, 3.14158, 3.14158, 3.14158, 3.14158, 0.08], dtype=np.float32)
rng = np.random.RandomState(2)
for t in range(int(os.environ.get("TICKS_STEPS", "30"))):
    s.step(rng.uniform(lo, hi, size=(N, 6)).astype(np.float32))
dev = s.dev
nc = torch.zeros(N, dtype=torch.int32, device=dev); cand = torch.zeros(N, 256, dtype=torch.int32, device=dev)
tk = torch.zeros(N, 256, dtype=torch.int32, device=dev); cr = torch.zeros(N, 256, 24, device=dev)
s.sim.debug_candidates(nc.data_ptr(), cand.data_ptr(), tk.data_ptr(), cr.data_ptr())
torch.cuda.synchronize()
nc = (nc.cpu().numpy() & 0xffff); cand = cand.cpu().numpy(); tk = tk.cpu().numpy(); cr = cr.cpu().numpy()
mask = np.arange(256)[None, :] < nc[:, None]
t = (tk[mask] & 0x0fffffff) * 1e-2          # us (the top four bits: contacts of the pair, tu_narrow.hip)
print("candidates", mask.sum(), "sum %.1f ms, mean %.2f us, p50 %.2f p90 %.2f p99 %.2f max %.1f" % (t.sum() * 1e-3, t.mean(), *np.percentile(t, [50, 90, 99]), t.max()))
print("sum/2048 waves = %.3f ms" % (t.sum() * 1e-3 / 2048))
ncon = (tk[mask].astype(np.uint32) >> 28).astype(np.int64); hit = ncon != 0
print("hit fraction %.2f; mean us hit %.2f miss %.2f" % (hit.mean(), t[hit].mean(), t[~hit].mean()))
c = cand[mask]; g1 = c & 0xffff; g2 = (c >> 16) & 0xffff
agg = collections.defaultdict(lambda: [0, 0.0, 0, 0.0, 0])
for a, b, x, h, k_ in zip(g1, g2, t, hit, ncon):
    k = (gn[a].split("/")[0] if "/" in gn[a] else gn[a], gn[b].split("/")[0] if "/" in gn[b] else gn[b])
    agg[k][0] += 1; agg[k][1] += x
    if h: agg[k][2] += 1; agg[k][3] += x; agg[k][4] += int(k_)
print("  (pair class: candidates, total time, mean; HITS: count, mean us, contacts per hit; MISSES: count, mean us)")
for k, (n, tot, nh, th, nk) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print("  %-50s n %6d  total %.1f ms  mean %.2f us | hits %6d %.2f us %.2f contacts | misses %6d %.2f us" % (k, n, tot * 1e-3, tot / n, nh, th / max(nh, 1), nk / max(nh, 1), n - nh, (tot - th) / max(n - nh, 1)))

ph = tk[:, 224:230].astype(np.float64)
tot = ph.sum(0)                      # accumulated over every substep of the run
whole = tot[0] + tot[5]
print("narrowphase time by phase (share of fetch + narrow_pair): fetch of the work item and both geoms %.1f %% | hull vertices into registers %.1f %% | "
      "flat-face scan incl. patch pass %.1f %% | MPR %.1f %% | contact output %.1f %%" % (100 * tot[0] / whole, 100 * tot[1] / whole, 100 * tot[2] / whole,
      100 * tot[3] / whole, 100 * (tot[5] - tot[1] - tot[2] - tot[3]) / whole))

fine = tk[:, 230:234].astype(np.float64).sum(0)
print("inside the face scan (share of fetch + narrow_pair): face setup %.1f %% | first support pass %.1f %% | outline tests %.1f %% | five-sample patch pass %.1f %%"
      % tuple(100 * fine / whole))

rp = tk[:, 234:237].astype(np.float64).sum(0)
if rp[1] > 0:
    print("row pass (light region, four pairs per wavefront): %.0f rows attempted, %.1f %% settled, %.2f us per pass-row (%.1f ms in all = %.1f %% of fetch + narrow_pair)"
          % (rp[1], 100 * rp[2] / rp[1], rp[0] * 1e-2 / rp[1], rp[0] * 1e-5, 100 * rp[0] / whole))

ex = tk[:, 237:240].astype(np.float64).sum(0)
print("after the query (share of fetch + narrow_pair): face_patch (five samples -> contacts, wave-uniform arithmetic) %.1f %% | hull-against-hull patches %.1f %%" % (100 * ex[0] / whole, 100 * ex[1] / whole))
