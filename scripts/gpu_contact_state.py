"""Prints the GPU contact list of one tests/golden/contact_rich_states.json state (debugging aid)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from so101_sim_amd.model import scenes
from tests.simharness import ArraySim

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 6
raw32, meta = scenes.load_blob("banana", "f32")
g = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "contact_rich_states.json")))
st = g["states"][idx]
Q = np.array(st["qpos"])[:, None]; V = np.array(st["qvel"])[:, None]; W = np.array(st["warm"])[:, None]; A = np.array(st["action"])[:, None]
sim = ArraySim(raw32, 1, backend=os.environ.get("BACKEND", "gpu"))
sim.set_state(Q, V, A, W)
d = sim.debug_forward()[0]
for c in d["contacts"]:
    print(c["geom1"], c["geom2"], "dist %.7f" % c["dist"], "pos", np.round(c["pos"], 6), "n", np.round(c["normal"], 5))
