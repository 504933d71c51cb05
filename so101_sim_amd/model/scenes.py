"""SO100 hand-over scene assembly: scene_pbr.xml + object prop + container prop + task constants.

Mirrors what `SO100HandOver.__init__` composes (so101_sim/tasks/so100_hand_over.py:128-236) on top
of `SO100Arena` (so101_sim/tasks/base/so100_task.py:386-395): the object prop is attached first,
then the container (:169, :199), container meshes are scaled (:187-192), overlap boxes come from
`SO100_HANDOVER_CONFIGS` (:80-118) and the reset distributions from :37-55.

Compiled blobs are committed under `blobs/` so that nothing at run time needs the MJCF/mesh assets;
`compile_scene` regenerates them when an asset tree is available (env `SO101_ASSETS`, or the
reference checkout in this container).
"""
from __future__ import annotations

import json
import os

import numpy as np

from . import blob as blobfmt
from . import mjcf

_HERE = os.path.dirname(os.path.abspath(__file__))
BLOB_DIR = os.path.join(_HERE, "blobs")

TABLE_HEIGHT = 0.4     # so100_hand_over.py:34
RESET_HEIGHT = 0.05    # so100_hand_over.py:35

# so100_hand_over.py:80-118 (numbers are task configuration data)
HANDOVER_CONFIGS = {
    "banana": dict(
        object_model="ycb/011_banana/google_64k/model.xml",
        container_model="ycb/024_bowl/google_64k/model.xml",
        container_mesh_scale=1.5, success_threshold=0.1,
        overlap_boxes=[dict(position=np.array([-0.017, -0.045, 0.035]) * 1.5,
                            half_extents=np.array([0.02, 0.02, 0.01]) * 1.5)],
        instruction="pick up the banana and put it in the bowl using the SO100 arm"),
    "pen": dict(
        object_model="edr/pen/model.xml",
        container_model="gso/BIA_Cordon_Bleu_White_Porcelain_Utensil_Holder_900028/model.xml",
        container_mesh_scale=0.6, success_threshold=0.05,
        overlap_boxes=[dict(position=np.array([0.0, 0.0, 0.02666]) * 0.6,
                            half_extents=np.array([0.04666, 0.04666, 0.025]) * 0.6),
                       dict(position=np.array([0.0, 0.0, 0.25]) * 0.6,
                            half_extents=np.array([0.1, 0.1, 0.01666]) * 0.6)],
        instruction="pick up the pen and put it in the container using the SO100 arm"),
}

SO100_HOME_CTRL = np.array([0.0, -1.57079, 1.57079, 1.57079, -1.57079, 0.0])   # so100_task.py:45-47


def find_assets() -> str | None:
    for cand in (os.environ.get("SO101_ASSETS"), "/root/reference/so101_sim/assets"):
        if cand and os.path.isdir(os.path.join(cand, "so100")):
            return cand
    return None


def time_limit_last_step(time_limit: float, control_timestep: float = 0.02, timestep: float = 0.002) -> int:
    """Index (1-based) of the control step on which `physics.time() >= time_limit` first holds,
    replaying MuJoCo's fp64 `time += timestep` accumulation."""
    nsub = int(round(control_timestep / timestep))
    t, step = 0.0, 0
    while True:
        step += 1
        for _ in range(nsub):
            t += timestep
        if t >= time_limit:
            return step


def compile_scene(object_name: str, assets: str | None = None) -> dict:
    assets = assets or find_assets()
    if assets is None:
        raise FileNotFoundError("MJCF assets not found: set SO101_ASSETS to .../so101_sim/assets")
    cfg = HANDOVER_CONFIGS[object_name]
    sc = mjcf.SceneCompiler()
    sc.add_scene(os.path.join(assets, "so100", "scene_pbr.xml"))
    obj = sc.add_free_prop(os.path.join(assets, cfg["object_model"]), "object")
    con = sc.add_free_prop(os.path.join(assets, cfg["container_model"]), "container",
                           mesh_scale=cfg["container_mesh_scale"])
    out = mjcf.finalize(sc)
    m = out["model"]
    nbox = len(cfg["overlap_boxes"])
    m.update(
        task_object_body=obj, task_container_body=con, task_nbox=nbox,
        task_box_pos=np.array([b["position"] for b in cfg["overlap_boxes"]]),
        task_box_half=np.array([b["half_extents"] for b in cfg["overlap_boxes"]]),
        task_dist_threshold=cfg["success_threshold"],
        task_obj_pos_lo=np.array([0.2, -0.1, TABLE_HEIGHT + RESET_HEIGHT]),
        task_obj_pos_hi=np.array([0.3, 0.1, TABLE_HEIGHT + RESET_HEIGHT]),
        task_obj_yaw=np.array([-np.pi * 0.1, np.pi * 0.1]),
        task_con_pos_lo=np.array([-0.3, -0.1, TABLE_HEIGHT + RESET_HEIGHT]),
        task_con_pos_hi=np.array([-0.2, 0.1, TABLE_HEIGHT + RESET_HEIGHT]),
        task_home_ctrl=SO100_HOME_CTRL,
    )
    out["meta"]["instruction"] = cfg["instruction"]
    out["meta"]["object_name"] = object_name
    return out


def blob_paths(object_name: str) -> tuple[str, str, str]:
    stem = os.path.join(BLOB_DIR, f"so100_handover_{object_name}")
    return stem + ".f32.bin", stem + ".f64.bin", stem + ".json"


def write_blobs(object_name: str, assets: str | None = None):
    out = compile_scene(object_name, assets)
    p32, p64, pj = blob_paths(object_name)
    os.makedirs(BLOB_DIR, exist_ok=True)
    with open(p32, "wb") as f:
        f.write(blobfmt.pack(out["model"], np.float32))
    with open(p64, "wb") as f:
        f.write(blobfmt.pack(out["model"], np.float64))
    with open(pj, "w") as f:
        json.dump(out["meta"], f, indent=1)
    return out


def load_blob(object_name: str, real: str = "f32") -> tuple[bytes, dict]:
    p32, p64, pj = blob_paths(object_name)
    path = p32 if real == "f32" else p64
    if not os.path.exists(path):
        write_blobs(object_name)
    with open(path, "rb") as f:
        raw = f.read()
    with open(pj) as f:
        meta = json.load(f)
    return raw, meta


if __name__ == "__main__":
    import sys
    for name in (sys.argv[1:] or ["banana", "pen"]):
        o = write_blobs(name)
        m = o["model"]
        print(name, {k: m[k] for k in ("nq", "nv", "nu", "nbody", "ngeom", "nvert", "npair")},
              "proxy:", o["meta"]["proxy_inertia"])


# ------------------------------------------------------------------------------------------------ ALOHA
# tasks/hand_over.py:33-56 (reset distributions), :80-118 (HANDOVER_CONFIGS: same boxes and scales as the SO100 task,
# other instructions), tasks/base/aloha2_task.py:38-45 (home pose), :57-70 (gripper ranges), :107 (table height offset)
ALOHA_TABLE_HEIGHT = 0.0
ALOHA_RESET_HEIGHT = 0.1
ALOHA_TABLE_HEIGHT_OFFSET = 0.011
ALOHA_HOME_CTRL = np.array([0.0, -0.96, 1.16, 0.0, -0.3, 0.0, 0.002])
ALOHA_HOME_QPOS = np.array([0.0, -0.959, 1.182, 0.0, -0.274, 0.0, 0.0082, 0.0082])
ALOHA_GRIPPER_LIMITS = dict(sim_qpos=(0.037, 0.0078), sim_ctrl=(0.037, 0.002), follower=(1.5155, -0.06135),
                            leader=(0.78, -0.04))      # (open, close)
ALOHA_INSTRUCTIONS = dict(banana="hand over the banana and put it in the bowl",
                          pen="hand over the pen and put it in the container")


def compile_aloha_scene(object_name: str | None, assets: str | None = None,
                        table_height_offset: float = ALOHA_TABLE_HEIGHT_OFFSET) -> dict:
    """aloha/scene_pbr.xml (+ the two props of a hand-over task when `object_name` is given; None = the bare AlohaTask
    of tasks/test/aloha2_task_test.py) as a general-tree model: two 8-dof arms (6 hinges + 2 slide fingers coupled by a
    joint equality), position actuators, joint damping, nq = 16 (+ 14), nv = 16 (+ 12), nu = 14."""
    assets = assets or find_assets()
    if assets is None:
        raise FileNotFoundError("MJCF assets not found: set SO101_ASSETS to .../so101_sim/assets")
    sc = mjcf.SceneCompiler()
    sc.general_tree = True
    sc.add_scene(os.path.join(assets, "aloha", "scene_pbr.xml"))
    if table_height_offset:            # aloha2_task.py:493-496 (the camera and the visual extrusions move too; no physics)
        for b in sc.bodies:
            if b.name == "table":
                b.pos = b.pos + np.array([0.0, 0.0, table_height_offset])
    obj = con = -1
    cfg = None
    if object_name is not None:
        cfg = HANDOVER_CONFIGS[object_name]
        obj = sc.add_free_prop(os.path.join(assets, cfg["object_model"]), "object")
        con = sc.add_free_prop(os.path.join(assets, cfg["container_model"]), "container",
                               mesh_scale=cfg["container_mesh_scale"])
    out = mjcf.finalize(sc)
    m = out["model"]
    # geom classes of the contact-sequence reward (hand_over.py:286-303): object, container, and the geoms below each arm's gripper_link
    names = out["meta"]["body_names"]
    parent = list(m["body_parent"])

    def below(body, root):
        while body != 0:
            if body == root:
                return True
            body = parent[body]
        return False
    lg, rg = names.index("left/gripper_link"), names.index("right/gripper_link")
    m["task_geom_class"] = np.array([(1 if b == obj else 0) | (2 if b == con else 0) | (4 if below(b, lg) else 0) | (8 if below(b, rg) else 0)
                                     for b in m["geom_body"]], np.int32)
    boxes = cfg["overlap_boxes"] if cfg else []
    z = ALOHA_TABLE_HEIGHT + ALOHA_RESET_HEIGHT
    m.update(
        task_object_body=obj, task_container_body=con, task_nbox=len(boxes),
        task_box_pos=np.array([b["position"] for b in boxes]).reshape(-1, 3),
        task_box_half=np.array([b["half_extents"] for b in boxes]).reshape(-1, 3),
        task_dist_threshold=cfg["success_threshold"] if cfg else 0.0,
        task_obj_pos_lo=np.array([0.12, -0.1, z]), task_obj_pos_hi=np.array([0.18, 0.1, z]),
        task_obj_yaw=np.array([-np.pi * 0.1 - np.pi * 0.5, np.pi * 0.1 - np.pi * 0.5]),
        task_con_pos_lo=np.array([-0.18, -0.1, z]), task_con_pos_hi=np.array([-0.12, 0.1, z]),
        task_home_ctrl=np.concatenate([ALOHA_HOME_CTRL, ALOHA_HOME_CTRL]),
        task_home_qpos=np.concatenate([ALOHA_HOME_QPOS, ALOHA_HOME_QPOS]),
        # observables and actions (aloha2_task.py:279-359,386-444): joints_pos = six arm joints and the LEFT finger of each arm,
        # the finger in follower units; joints_vel = all sixteen joint velocities; action = ctrl with the grippers in follower units
        task_obs_qposadr=np.array([0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14], np.int32),
        task_obs_is_gripper=np.array([0] * 6 + [1] + [0] * 6 + [1], np.int32),
        task_act_is_gripper=np.array([0] * 6 + [1] + [0] * 6 + [1], np.int32),
        task_gripper_limits=np.array([ALOHA_GRIPPER_LIMITS["sim_qpos"][0], ALOHA_GRIPPER_LIMITS["sim_qpos"][1],
                                      ALOHA_GRIPPER_LIMITS["sim_ctrl"][0], ALOHA_GRIPPER_LIMITS["sim_ctrl"][1],
                                      ALOHA_GRIPPER_LIMITS["follower"][0], ALOHA_GRIPPER_LIMITS["follower"][1]]),
    )
    out["meta"]["instruction"] = ALOHA_INSTRUCTIONS.get(object_name, "")
    out["meta"]["object_name"] = object_name
    out["meta"]["keyframes"] = {k: {f: v.tolist() for f, v in d.items()} for k, d in sc.keyframes.items()}
    return out


def aloha_blob_paths(object_name: str | None) -> tuple[str, str, str]:
    stem = os.path.join(BLOB_DIR, f"aloha_handover_{object_name}" if object_name else "aloha_bare")
    return stem + ".f32.bin", stem + ".f64.bin", stem + ".json"


def write_aloha_blobs(object_name: str | None, assets: str | None = None):
    out = compile_aloha_scene(object_name, assets)
    p32, p64, pj = aloha_blob_paths(object_name)
    os.makedirs(BLOB_DIR, exist_ok=True)
    with open(p32, "wb") as f:
        f.write(blobfmt.pack(out["model"], np.float32))
    with open(p64, "wb") as f:
        f.write(blobfmt.pack(out["model"], np.float64))
    with open(pj, "w") as f:
        json.dump(out["meta"], f, indent=1)
    return out


def load_aloha_blob(object_name: str | None, real: str = "f64") -> tuple[bytes, dict]:
    p32, p64, pj = aloha_blob_paths(object_name)
    path = p32 if real == "f32" else p64
    if not os.path.exists(path):
        write_aloha_blobs(object_name)
    with open(path, "rb") as f:
        raw = f.read()
    with open(pj) as f:
        meta = json.load(f)
    return raw, meta


# ------------------------------------------------------------------------------------------------ Dining
# tasks/base/dining.py:39-160 (the scene: the ALOHA robot and SIX free props, attached in the order mug, pen, banana, plate, bowl,
# container; plate meshes scaled 0.8, bowl 1.5, container 0.6), :162-228 (six table regions, the three top and the three bottom ones
# shuffled among (plate, bowl, container) and (mug, pen, banana)), :230-267 (PropPlacer: positions in the order plate, bowl, container,
# mug, pen, banana, a uniform yaw each, collisions ignored, then settled); tasks/dining_place_in_container.py:26-160 (three tasks on
# that scene: object, receptacle, reward type, overlap boxes, instruction).
DINING_TABLE_HEIGHT = 0.03
DINING_RESET_HEIGHT = 0.03
DINING_PROPS = (      # attach order (= body order), model, mesh scale
    ("mug", "ycb/025_mug/google_64k/model.xml", None),
    ("pen", "edr/pen/model.xml", None),
    ("banana", "ycb/011_banana/google_64k/model.xml", None),
    ("plate", "ycb/029_plate/google_64k/model.xml", 0.8),
    ("bowl", "ycb/024_bowl/google_64k/model.xml", 1.5),
    ("container", "gso/BIA_Cordon_Bleu_White_Porcelain_Utensil_Holder_900028/model.xml", 0.6),
)
DINING_PLACER_ORDER = ("plate", "bowl", "container", "mug", "pen", "banana")       # dining.py:234-251
_Z = DINING_TABLE_HEIGHT + DINING_RESET_HEIGHT
DINING_REGIONS = np.array([      # (low, high) of top left / middle / right, bottom left / middle / right (dining.py:163-186)
    [[-0.3, 0.1, _Z], [-0.22, 0.2, _Z]], [[-0.03, 0.1, _Z], [0.03, 0.2, _Z]], [[0.22, 0.1, _Z], [0.3, 0.2, _Z]],
    [[-0.3, -0.25, _Z], [-0.2, -0.1, _Z]], [[-0.05, -0.25, _Z], [0.05, -0.1, _Z]], [[0.2, -0.25, _Z], [0.3, -0.1, _Z]]])
DINING_TASKS = {                  # dining_place_in_container.py:35-80
    "banana": dict(object="banana", receptacle="bowl", reward="bbox", instruction="put the banana in the bowl",
                   overlap_boxes=[dict(position=np.array([-0.017, -0.045, 0.035]) * 1.5, half_extents=np.array([0.02, 0.02, 0.01]) * 1.5)]),
    "pen": dict(object="pen", receptacle="container", reward="bbox", instruction="put the pen in the white cup",
                overlap_boxes=[dict(position=np.array([0.0, 0.0, 0.02666]) * 0.6, half_extents=np.array([0.04666, 0.04666, 0.025]) * 0.6),
                               dict(position=np.array([0.0, 0.0, 0.25]) * 0.6, half_extents=np.array([0.1, 0.1, 0.01666]) * 0.6)]),
    "mug": dict(object="mug", receptacle="plate", reward="contact", instruction="put the red mug on the plate", overlap_boxes=[]),
}
DINING_REWARD_MODE = dict(bbox=0, contact=2)       # so101_tree_config.reward_mode


def dining_task_entries(model: dict, meta: dict, task_id: str) -> dict:
    """The task_* entries of one Dining task on the compiled scene (the scene itself is the same for the three tasks): object and
    receptacle bodies, overlap boxes, and the geom classes of the contact reward (1 object, 2 receptacle)."""
    cfg = DINING_TASKS[task_id]
    names = meta["body_names"]
    obj, con = names.index(cfg["object"]), names.index(cfg["receptacle"])
    boxes = cfg["overlap_boxes"]
    return dict(
        task_object_body=obj, task_container_body=con, task_nbox=len(boxes),
        task_box_pos=np.array([b["position"] for b in boxes]).reshape(-1, 3), task_box_half=np.array([b["half_extents"] for b in boxes]).reshape(-1, 3),
        task_geom_class=np.array([(1 if b == obj else 0) | (2 if b == con else 0) for b in model["geom_body"]], np.int32))


def compile_dining_scene(task_id: str = "banana", assets: str | None = None, table_height_offset: float = ALOHA_TABLE_HEIGHT_OFFSET) -> dict:
    """aloha/scene_pbr.xml + the six props of tasks/base/dining.py as a general-tree model: nq = 16 + 42 = 58, nv = 16 + 36 = 52, nu = 14,
    28 bodies, 240 collision geoms.  One scene serves DiningPlaceBananaInBowl / PenInContainer / MugOnPlate; `task_id` selects the task
    entries written into the blob (so101_sim_amd.aloha re-packs them for the other two)."""
    assets = assets or find_assets()
    if assets is None:
        raise FileNotFoundError("MJCF assets not found: set SO101_ASSETS to .../so101_sim/assets")
    sc = mjcf.SceneCompiler()
    sc.general_tree = True
    sc.add_scene(os.path.join(assets, "aloha", "scene_pbr.xml"))
    if table_height_offset:
        for b in sc.bodies:
            if b.name == "table":
                b.pos = b.pos + np.array([0.0, 0.0, table_height_offset])
    ids = {}
    for name, path, scale in DINING_PROPS:
        ids[name] = sc.add_free_prop(os.path.join(assets, path), name, **({} if scale is None else dict(mesh_scale=scale)))
    out = mjcf.finalize(sc)
    m = out["model"]
    m.update(dining_task_entries(m, out["meta"], task_id))
    m.update(
        task_kind=1, task_dist_threshold=0.0,
        task_prop_bodies=np.array([ids[n] for n in DINING_PLACER_ORDER], np.int32),
        task_region_lo=DINING_REGIONS[:, 0, :].copy(), task_region_hi=DINING_REGIONS[:, 1, :].copy(),
        # (the hand-over placement entries are unused by task_kind 1; the yaw range is the props' uniform_z_rotation, dining.py:33-36)
        task_obj_pos_lo=np.zeros(3), task_obj_pos_hi=np.zeros(3), task_obj_yaw=np.array([-np.pi, np.pi]),
        task_con_pos_lo=np.zeros(3), task_con_pos_hi=np.zeros(3),
        task_home_ctrl=np.concatenate([ALOHA_HOME_CTRL, ALOHA_HOME_CTRL]), task_home_qpos=np.concatenate([ALOHA_HOME_QPOS, ALOHA_HOME_QPOS]),
        task_obs_qposadr=np.array([0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14], np.int32),
        task_obs_is_gripper=np.array([0] * 6 + [1] + [0] * 6 + [1], np.int32), task_act_is_gripper=np.array([0] * 6 + [1] + [0] * 6 + [1], np.int32),
        task_gripper_limits=np.array([ALOHA_GRIPPER_LIMITS["sim_qpos"][0], ALOHA_GRIPPER_LIMITS["sim_qpos"][1], ALOHA_GRIPPER_LIMITS["sim_ctrl"][0],
                                      ALOHA_GRIPPER_LIMITS["sim_ctrl"][1], ALOHA_GRIPPER_LIMITS["follower"][0], ALOHA_GRIPPER_LIMITS["follower"][1]]),
    )
    out["meta"]["instruction"] = DINING_TASKS[task_id]["instruction"]
    out["meta"]["task_id"] = task_id
    out["meta"]["prop_bodies"] = ids
    out["meta"]["keyframes"] = {k: {f: v.tolist() for f, v in d.items()} for k, d in sc.keyframes.items()}
    return out


def dining_blob_paths() -> tuple[str, str, str]:
    stem = os.path.join(BLOB_DIR, "aloha_dining")
    return stem + ".f32.bin", stem + ".f64.bin", stem + ".json"


def write_dining_blobs(assets: str | None = None):
    out = compile_dining_scene("banana", assets)
    p32, p64, pj = dining_blob_paths()
    os.makedirs(BLOB_DIR, exist_ok=True)
    with open(p32, "wb") as f:
        f.write(blobfmt.pack(out["model"], np.float32))
    with open(p64, "wb") as f:
        f.write(blobfmt.pack(out["model"], np.float64))
    with open(pj, "w") as f:
        json.dump(out["meta"], f, indent=1)
    return out


def load_dining_blob(task_id: str = "banana", real: str = "f64") -> tuple[bytes, dict]:
    """The Dining scene with the task entries of `task_id` (banana | pen | mug): the committed blob carries those of "banana"; for the
    other two the entries are replaced and the blob packed again (same scene, same numbers)."""
    if task_id not in DINING_TASKS:
        raise ValueError(f"unknown Dining task {task_id!r}, must be one of {list(DINING_TASKS)}")
    p32, p64, pj = dining_blob_paths()
    path = p32 if real == "f32" else p64
    if not os.path.exists(path):
        write_dining_blobs()
    with open(path, "rb") as f:
        raw = f.read()
    with open(pj) as f:
        meta = json.load(f)
    if task_id != "banana":
        m = blobfmt.unpack(raw)
        m.update(dining_task_entries(m, meta, task_id))
        raw = blobfmt.pack(m, np.float32 if real == "f32" else np.float64)
        meta = dict(meta, instruction=DINING_TASKS[task_id]["instruction"], task_id=task_id)
    return raw, meta
