"""Transcribes the known-answer vectors printed in the reference's executed notebooks into
tests/golden/kat*.json.  Runs only where the reference checkout is mounted (this container); the JSON
outputs are committed.  Source cells: so101_rl.ipynb (KAT-1), examples/so101_rl_breakdown.ipynb (KAT-2/3)."""
import json
import os
import re
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
KEYS = ["commanded_joints_pos", "joints_pos", "joints_vel", "physics_state", "undelayed_joints_pos",
        "undelayed_joints_vel", "delayed_physics_state"]


def cell_texts(path):
    nb = json.load(open(path))
    for ci, c in enumerate(nb["cells"]):
        if c["cell_type"] != "code":
            continue
        text = "".join("".join(o.get("text", [])) for o in c.get("outputs", []) if "text" in o)
        yield ci, "".join(c["source"]), text


def arrays(text):
    out = {}
    for k in KEYS:
        m = re.search(r"'" + k + r"': array\(\[(.*?)\](?:, dtype=\w+)?\)", text, re.S)
        if m:
            body = m.group(1).replace("\n", " ")
            out[k] = [float(x) for x in body.split(",") if x.strip()]
    return out


def main():
    # KAT-1
    for ci, src, text in cell_texts(os.path.join(REF, "so101_rl.ipynb")):
        if "env.step(np.array(gripper_open))" in src and "physics_state" in text:
            a = arrays(text)
            rew = float(re.search(r"Reward: ([-\d.e]+)", text).group(1))
            disc = float(re.search(r"Discount: ([-\d.e]+)", text).group(1))
            kat1 = dict(source="so101_rl.ipynb cell %d: reset(); step([0,0,0,0,0,0.5]); calibration file found (CWD = repo root)" % ci,
                        action=[0, 0, 0, 0, 0, 0.5], observation=a, reward=rew, discount=disc)
            json.dump(kat1, open(os.path.join(OUT, "kat1.json"), "w"), indent=1)
            print("kat1", {k: len(v) for k, v in a.items()}, rew, disc)
    # KAT-2 / KAT-3: examples/so101_rl_breakdown.ipynb is not valid JSON in the reference checkout
    # (a cell is missing a comma), so it is scanned as raw text.
    raw = open(os.path.join(REF, "examples", "so101_rl_breakdown.ipynb")).read()

    def raw_array(key, start=0):
        m = re.compile(r"'" + key + r"': array\(\[(.*?)\]", re.S).search(raw, start)
        body = re.sub(r"[^0-9eE+\-., ]", " ", m.group(1).replace("\\n", " "))
        return [float(x) for x in body.split(",") if x.strip()], m.end()

    first = raw.index("'commanded_joints_pos': array")
    obs = {}
    for k in KEYS:
        try:
            obs[k], _ = raw_array(k, first - 10)
        except Exception:
            obs[k] = []
    kat2 = dict(source="examples/so101_rl_breakdown.ipynb (first printed observation dict): state right after reset(); "
                       "calibration file NOT found (CWD = examples/)", observation=obs)
    json.dump(kat2, open(os.path.join(OUT, "kat2.json"), "w"), indent=1)
    print("kat2", {k: len(v) for k, v in obs.items()})
    kat3 = {}
    m = re.search(r"velocity=([\d.e-]+) and acceleration=([\d.e-]+)", raw)
    kat3["settle_warning"] = dict(velocity=float(m.group(1).rstrip(".")), acceleration=float(m.group(2).rstrip(".")), attempts=1, seconds=2.0)
    m = re.search(r"Observation keys: \[(.*?)\]", raw)
    kat3["obs_keys"] = re.findall(r"'(\w+)'", m.group(1))
    kat3["first_timestep"] = dict(reward=None, discount=None)
    kat3["action_shape"] = [int(re.search(r"Action shape: \((\d+),\)", raw).group(1))]
    kat3["action_dtype"] = re.search(r"Action dtype: (\w+)", raw).group(1)
    kat3["action_ranges_2dp"] = [[float(a), float(b)] for a, b in re.findall(r"range: \[\s*([-\d.]+),\s*([-\d.]+)\]", raw)[:6]]
    steps = re.findall(r"Step\s+\d+ \| Action: \[(.*?)\] \| Reward: ([-\d.]+)", raw)
    kat3["random_steps"] = [dict(action=[float(x) for x in a.split()], reward=float(r)) for a, r in steps[:5]]
    json.dump(kat3, open(os.path.join(OUT, "kat3.json"), "w"), indent=1)
    print("kat3", json.dumps(kat3)[:600])


if __name__ == "__main__":
    main()
