"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports everything
include/so101.h declares, rejects bad blobs without touching a GPU, and the Python registry/factory
behaves like so101_sim/task_suite.py."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(hip_lib):
    from so101_sim_amd import native
    header = open(os.path.join(ROOT, "include", "so101.h")).read()
    declared = set(re.findall(r"\b(so101_[a-z_]+)\s*\(", header))
    assert declared == set(native.EXPORTS)
    lib = native.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.so101_version() == 10 and lib.so101_max_contacts() >= 16


def test_create_rejects_bad_blobs_without_gpu(hip_lib, blobs):
    from so101_sim_amd import native
    with pytest.raises(RuntimeError, match="bad blob magic"):
        native.Sim(b"\0" * 64, 4)
    with pytest.raises(RuntimeError, match="f32 blob"):
        native.Sim(blobs["f64"], 4)
    with pytest.raises(RuntimeError, match="truncated"):
        native.Sim(blobs["f32"][:4000], 4)
    lib = native.load_library()
    assert lib.so101_configure(None, None) == -1 and lib.so101_step(None, None, None, None, None, None, None) == -1


def test_missing_library_fails_loudly(tmp_path):
    from so101_sim_amd import native
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        native.load_library(str(tmp_path / "nope.so"))


def test_product_never_imports_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "so101_sim_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(base, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", text, re.M) or "so101_oracle" in text or "oracle/" in text:
                    bad.append(f)
    assert not bad, bad


def test_registry_matches_reference_names():
    from so101_sim_amd import task_suite
    keys = list(task_suite.TASK_FACTORIES.keys())
    assert len(keys) == 22
    assert keys[:3] == ["BlocksSpelling", "BowlOnRack", "DesktopWrapHeadphone"] and keys[-2:] == ["SO100HandOverPen", "SO100HandOverBanana"]
    assert task_suite.TASK_FACTORIES["SO100HandOverBanana"][1] == {"object_name": "banana"}
    assert task_suite.DEFAULT_CONTROL_TIMESTEP == 0.02
    assert task_suite.DEFAULT_CAMERAS == ("overhead_cam", "worms_eye_cam", "wrist_cam_left", "wrist_cam_right")


def test_reference_import_path_works_without_an_install_call():
    """`from so101_sim import task_suite` (run_eval.py:22, scripts/so101_lerobot_wrapper.py:12, so101_rl.ipynb) resolves to this build from a fresh
    interpreter with the repo root on the path - no install_as_so101_sim() line - and hands out the SAME registry and factory objects."""
    import subprocess
    import sys
    code = ("from so101_sim import task_suite\n"
            "import so101_sim, so101_sim_amd.task_suite as t\n"
            "assert task_suite.create_task_env is t.create_task_env and task_suite.TASK_FACTORIES is t.TASK_FACTORIES\n"
            "assert so101_sim.task_suite is task_suite and len(task_suite.TASK_FACTORIES) == 22\n"
            "assert task_suite.DEFAULT_CONTROL_TIMESTEP == 0.02 and task_suite.SO100HandOver is t.SO100HandOver\n"
            "try:\n    task_suite.create_task_env('Nope', time_limit=1.0)\nexcept ValueError as e:\n    assert 'Unknown task_name: Nope' in str(e)\nelse:\n    raise SystemExit(3)\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stderr
    # nothing of the reference lives in the alias package: two short files that import so101_sim_amd
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "so101_sim")) if not f.startswith("__pycache__"))
    assert files == ["__init__.py", "task_suite.py"]
    assert all(len(open(os.path.join(ROOT, "so101_sim", f)).read().splitlines()) < 20 for f in files)


def test_factory_errors_like_reference():
    from so101_sim_amd import task_suite
    with pytest.raises(ValueError, match="Unknown task_name: Nope. Available tasks:"):
        task_suite.create_task_env("Nope", time_limit=1.0)
    with pytest.raises(NotImplementedError):
        task_suite.create_task_env("BowlOnRack", time_limit=1.0)              # ALOHA tasks other than the hand-over: registry key exists, not built
    assert task_suite.HandOver(object_name="banana", reward_based_on_overlap=False, reward_requires_handover=True).reward_requires_handover
    with pytest.raises(ValueError, match="Invalid object name"):
        task_suite.HandOver(object_name="mug")
    t = task_suite.HandOver(object_name="pen", control_timestep=0.02, cameras=())
    assert t.get_instruction() == "hand over the pen and put it in the container"       # hand_over.py:117
    with pytest.raises(ValueError, match="Invalid object name"):
        task_suite.SO100HandOver(object_name="mug")


def test_task_kwargs_and_calibration_cwd_quirk(tmp_path, monkeypatch):
    from so101_sim_amd import task_suite
    monkeypatch.chdir(tmp_path)
    t = task_suite.SO100HandOver(object_name="banana", control_timestep=0.02, cameras=())
    assert np.all(t.calibration.homing_offsets == 0)                            # file not found -> zeros, silently
    assert t.get_instruction() == "pick up the banana and put it in the bowl using the SO100 arm"
    os.makedirs(tmp_path / "calibration")
    (tmp_path / "calibration" / "red_arm.json").write_text(
        '{"shoulder_pan": {"homing_offset": 28}, "shoulder_lift": {"homing_offset": 42}, "elbow_flex": {"homing_offset": 18},'
        ' "wrist_flex": {"homing_offset": -21}, "wrist_roll": {"homing_offset": 1009}, "gripper": {"homing_offset": -158}}')
    t = task_suite.SO100HandOver(object_name="banana")
    assert list(t.calibration.homing_offsets) == [28, 42, 18, -21, 1009, -158]
    with pytest.raises(ValueError, match="Expected 6 joint positions, got 5"):
        t.calibration.apply_calibration_to_action(np.zeros(5))


def test_dmenv_lookalikes():
    from so101_sim_amd._dmenv import StepType, TimeStep
    ts = TimeStep(StepType.FIRST, None, None, {})
    assert ts.first() and not ts.mid() and not ts.last() and ts.reward is None and ts.discount is None
    assert TimeStep(StepType.LAST, 1.0, 0.0, {}).last()


def test_shard_ranges_partition_the_envs():
    from so101_sim_amd.distributed import shard_range
    for n, w in ((262144, 8), (4096, 2), (10, 3)):
        spans = [shard_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_lerobot_packaging_format():
    """so101_lerobot_wrapper.py:77-122: keys, dtypes, the 0.1 s timestamp quirk, zero action on reset; batched shapes."""
    import numpy as np
    import torch
    from so101_sim_amd import lerobot
    o = lerobot.to_lerobot_format(np.arange(6, dtype=np.float64), None, 0, 3)
    assert set(o) == {"observation.state", "action", "timestamp", "frame_index", "episode_index", "index", "task_index", "task"}
    assert o["observation.state"].dtype == torch.float32 and o["observation.state"].shape == (6,)
    assert torch.equal(o["action"], torch.zeros(6)) and o["timestamp"].item() == 0.0 and o["episode_index"].item() == 3
    o = lerobot.to_lerobot_format(np.zeros(6), np.ones(6), 7, 0)
    assert abs(o["timestamp"].item() - 0.7) < 1e-6 and o["frame_index"].item() == 7 and o["index"].item() == 7
    assert o["frame_index"].dtype == torch.long and o["task"] == "SO100 manipulation task" and o["task_index"].item() == 0
    b = lerobot.to_lerobot_format(np.zeros((5, 6)), None, 2, 1, n_envs=5)
    assert b["observation.state"].shape == (5, 6) and b["action"].shape == (5, 6) and b["frame_index"].shape == (5,)
    assert torch.all(b["frame_index"] == 2) and torch.allclose(b["timestamp"], torch.full((5,), 0.2))


def test_lerobot_packaging_matches_the_reference_wrapper():
    """tests/golden/lerobot_cases.json holds outputs of the reference's own SO101LeRobotWrapper._convert_to_lerobot_format
    (scripts/make_golden_lerobot.py executes it on synthetic time steps): same keys, dtypes, shapes and bit-identical
    values from so101_sim_amd.lerobot.to_lerobot_format."""
    import json
    import numpy as np
    import torch
    from so101_sim_amd.lerobot import to_lerobot_format
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "lerobot_cases.json")))
    assert len(g["cases"]) >= 10
    for c in g["cases"]:
        act = None if c["action"] is None else np.asarray(c["action"], dtype=c["action_dtype"])
        out = to_lerobot_format(np.asarray(c["joints_pos"]), act, c["frame_index"], c["episode_index"], device="cpu", n_envs=1)
        assert set(out) == set(c["expected"]), (set(out) ^ set(c["expected"]))
        for k, want in c["expected"].items():
            if isinstance(want, str):
                assert out[k] == want
                continue
            got = out[k]
            assert str(got.dtype) == want["dtype"] and list(got.shape) == want["shape"], (k, got.dtype, got.shape, want)
            assert got.double().flatten().tolist() == want["value"], k


def test_settled_cache_file_format(tmp_path):
    """Host logic of the on-disk settled-state store: round trip, key check, damage detection."""
    from so101_sim_amd import settled_cache as sc
    E, N = 3, 5
    rng = np.random.RandomState(0)
    arrays = {"qpos": rng.randn(E, 20, N).astype(np.float32), "qvel": rng.randn(E, 18, N).astype(np.float32),
              "warmstart": rng.randn(E, 18, N).astype(np.float32), "flags": rng.randint(0, 64, size=(E, N)).astype(np.int32)}
    key = {"seed": 7, "blob_sha256": "ab", "env_id_base": 4096, "action_offset": [0.0] * 6, "build": "x"}
    path = str(tmp_path / "s.bin")
    sc.write(path, key, 2, arrays)
    header, got = sc.read(path, expect_key=dict(key))
    assert header["first_episode"] == 2 and header["n_episodes"] == E and header["n_envs"] == N
    for k in arrays:
        np.testing.assert_array_equal(got[k], arrays[k])
        assert got[k].dtype == arrays[k].dtype
    with pytest.raises(sc.SettledCacheError, match="env_id_base"):          # another shard's file
        sc.read(path, expect_key=dict(key, env_id_base=0))
    with pytest.raises(sc.SettledCacheError, match="build"):
        sc.read(path, expect_key=dict(key, build="y"))
    data = bytearray(open(path, "rb").read())
    data[len(data) // 2] ^= 1
    open(path, "wb").write(bytes(data))
    with pytest.raises(sc.SettledCacheError, match="checksum"):
        sc.read(path)
    open(path, "wb").write(bytes(data[:100]))
    with pytest.raises(sc.SettledCacheError):
        sc.read(path)
    with pytest.raises(sc.SettledCacheError, match="shape"):
        sc.write(path, key, 0, dict(arrays, qpos=arrays["qpos"][:, :19]))


def test_aloha_action_spec():
    """AlohaTask.action_spec (aloha2_task.py:279-301) from the compiled model's ctrlrange."""
    from so101_sim_amd import aloha
    from so101_sim_amd.model import blob as blobfmt, scenes
    m = blobfmt.unpack(scenes.load_aloha_blob("banana", "f32")[0])
    spec = aloha.aloha_action_spec(np.asarray(m["act_ctrlrange"], dtype=np.float64).reshape(-1, 2))
    assert spec.shape == (14,) and spec.dtype == np.float32
    assert spec.minimum[0] == spec.minimum[7] == np.float32(-np.pi / 2) and spec.maximum[0] == np.float32(np.pi / 2)
    assert spec.minimum[6] == spec.minimum[13] == np.float32(-0.06135) and spec.maximum[6] == spec.maximum[13] == np.float32(1.5155)
    np.testing.assert_allclose(spec.minimum[1:6], [-1.85005, -1.76278, -3.14158, -1.8675, -3.14158], rtol=1e-6)
    np.testing.assert_allclose(spec.maximum[8:13], [1.25664, 1.6057, 3.14158, 2.23402, 3.14158], rtol=1e-6)
