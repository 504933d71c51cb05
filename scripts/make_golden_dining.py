"""Generates tests/golden/dining_placements.json by EXECUTING the reference's own `Dining._sample_props`
(/root/reference/so101_sim/tasks/base/dining.py:162-228, loaded from the reference checkout - never copied): the module imports
dm_control, which is not installable here, so the ONE method is taken out of the parsed file with `ast` (together with the two
module constants it reads, _TABLE_HEIGHT and _RESET_HEIGHT) and compiled on its own; it touches nothing but its `random_state`
argument (numpy) and those constants.  For each seed the fixture holds the method's output (prop name -> position) for
np.random.RandomState(seed), and the six `uniform(-pi, pi)` draws that follow from the same generator - the yaws dm_control's
PropPlacer draws next, one per prop in placer order (dining.py:234-251; third-party behaviour, restated, not executed)."""
import ast
import json
import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
SRC = os.path.join(REF, "so101_sim", "tasks", "base", "dining.py")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dining_placements.json")


def main():
    tree = ast.parse(open(SRC).read())
    keep = []
    for node in tree.body:
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id in ("_TABLE_HEIGHT", "_RESET_HEIGHT") for t in node.targets):
            keep.append(node)
        if isinstance(node, ast.ClassDef) and node.name == "Dining":
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name == "_sample_props":
                    keep.append(item)
    assert len(keep) == 3, [type(k).__name__ for k in keep]
    ns = {}
    exec(compile(ast.Module(body=keep, type_ignores=[]), SRC, "exec"), ns)
    sample = ns["_sample_props"]
    cases = []
    for seed in (0, 1, 2, 3, 7, 123, 2024, 99991):
        rs = np.random.RandomState(seed)
        pos = sample(None, rs)
        yaws = [float(rs.uniform(-np.pi, np.pi)) for _ in range(6)]
        cases.append(dict(seed=seed, positions={k: [float(x) for x in v] for k, v in pos.items()}, yaws_plate_bowl_container_mug_pen_banana=yaws))
    json.dump(dict(source="so101_sim/tasks/base/dining.py:162-228 (Dining._sample_props) executed on np.random.RandomState(seed); yaws: the six uniform(-pi, pi) draws that follow",
                   table_height=ns["_TABLE_HEIGHT"], reset_height=ns["_RESET_HEIGHT"], cases=cases), open(OUT, "w"), indent=1)
    print("wrote", OUT, len(cases), "cases")


if __name__ == "__main__":
    main()
