"""Per-state report of the contact-rich fixture on the GPU: contact-list agreement with the fp64 oracle (tests/parity_cases.py)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import parity_cases as pc
from tests.simharness import ArraySim
from so101_sim_amd.model import scenes
from oracle.oracle import Oracle

raw32, _ = scenes.load_blob("banana", "f32"); raw64, _ = scenes.load_blob("banana", "f64")
states = json.load(open(os.path.join(ROOT, "tests", "golden", "contact_rich_states.json")))["states"]
n = len(states)
Q = np.array([s["qpos"] for s in states]).T; V = np.array([s["qvel"] for s in states]).T
W = np.array([s["warm"] for s in states]).T; A = np.array([s["action"] for s in states]).T
sim = ArraySim(raw32, n, backend=sys.argv[1] if len(sys.argv) > 1 else "gpu")
sim.set_state(Q, V, A, W)
dbg = sim.debug_forward()
for e in range(n):
    o = Oracle(raw64); o.set_state(Q[:, e], V[:, e], W[:, e]); o.set_ctrl(A[:, e]); o.forward()
    a, ref = o.qacc()[0], o.contacts()
    problems, t, l, w = pc._compare_contact_lists(dbg[e]["contacts"], ref)
    err = np.abs(dbg[e]["qacc"] - a).max() / np.abs(a).max()
    o.inject_contacts(dbg[e]["contacts"]); o.forward(); a2 = o.qacc()[0]
    err2 = np.abs(dbg[e]["qacc"] - a2).max() / np.abs(a2).max()
    print(e, "ncon", len(ref), "overflow", dbg[e]["overflow"], "loose", l, "witness", w, "qacc err own %.2e injected %.2e" % (err, err2), "max|qvel| %.1f" % np.abs(V[:, e]).max(), problems)
