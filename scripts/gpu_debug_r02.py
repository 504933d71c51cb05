import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from so101_sim_amd.model import scenes
from oracle.oracle import Oracle
from tests.simharness import ArraySim
raw32, _ = scenes.load_blob("banana", "f32"); raw64, _ = scenes.load_blob("banana", "f64")
golden = json.load(open("tests/golden/contact_rich_states.json"))
states = golden["states"][:2]
n = len(states)
Q = np.array([s["qpos"] for s in states]).T; V = np.array([s["qvel"] for s in states]).T
W = np.array([s["warm"] for s in states]).T; A = np.array([s["action"] for s in states]).T
sim = ArraySim(raw32, n, backend="gpu")
sim.set_state(Q, V, A, W)
dbg = sim.debug_forward()
for e in range(n):
    o = Oracle(raw64); o.set_state(Q[:, e], V[:, e], W[:, e]); o.set_ctrl(A[:, e]); o.forward()
    ref = o.contacts(); mine = dbg[e]["contacts"]
    print("env", e, "ncon gpu", len(mine), "oracle", len(ref), "overflow", dbg[e]["overflow"], "ncand", dbg[e]["ncand"])
    for k in range(max(len(mine), len(ref))):
        a = mine[k] if k < len(mine) else None; b = ref[k] if k < len(ref) else None
        fa = "(%2d,%2d) %9.6f %s" % (a["geom1"], a["geom2"], a["dist"], np.round(a["pos"], 4)) if a else "-"
        fb = "(%2d,%2d) %9.6f %s" % (b["geom1"], b["geom2"], b["dist"], np.round(b["pos"], 4)) if b else "-"
        print("   ", fa, " | ", fb)
    a_o, _ = o.qacc()
    print(" qacc err", np.abs(dbg[e]["qacc"] - a_o).max() / np.abs(a_o).max(), "iters", dbg[e]["iters"])
