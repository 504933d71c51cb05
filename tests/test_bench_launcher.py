"""`python bench.py --gpus 2` launched PLAINLY (no torchrun environment) must start two ranks itself: the real bench.py
main runs here on CPU with gloo and a stub env factory (tests/bench_stub.py) in place of the GPU env."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--device", "cpu", "--env-factory", "tests.bench_stub:make",
                           "--no-cpu-baseline", *extra], env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_gpus_2_spawns_two_ranks_and_reports_the_whole_job():
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--envs-per-gpu", "8", "--repeats", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(line) == 1, r.stdout
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1
    assert out["config"]["global_envs"] == 16 and out["config"]["envs_per_gpu"] == 8
    # MAX over ranks: rank 1 sleeps 30 ms per step, rank 0 10 ms; the job time is rank 1's
    assert out["ms_per_step"] >= 30.0 * 0.9
    assert abs(out["value"] - 16 * 4 / (out["ms_per_step"] * 4e-3)) < 1e-6 * out["value"]
    # `value` is the mean over the windows (total steps / total time); the first window stays in the line as a field
    assert out["repeats"]["n"] == 2 and out["repeats"]["values"][0] == out["first_window"]["value"]
    tot = sum(16 * 4 / v for v in out["repeats"]["values"])
    assert abs(out["value"] - 2 * 16 * 4 / tot) < 1e-6 * out["value"]
    # evidence from the initialised process group itself (not from the environment): backend, world size, the ranks that answered
    # an all-gather, every rank's own ms per step (rank 0 sleeps 10 ms per step, rank 1 30 ms)
    d = out["dist"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and d["ranks_reporting"] == 2 and d["ranks"] == [0, 1]
    assert len(d["per_rank"]) == 2 and 10.0 * 0.9 <= d["per_rank"][0] < d["per_rank"][1] and d["per_rank"][1] >= 30.0 * 0.9
    # returns gathered in rank order over global env ids 0..15: (warm-up 1 + 2 windows x 4 steps) x id; mean id = 7.5
    assert abs(out["mean_episode_return"] - 9 * 7.5) < 1e-4


def test_periodic_all_gather_of_returns_inside_a_long_window():
    """SURVEY 8e / BASELINE configs[4]: every 100 control steps the ranks all-gather their episode returns (logging only, on a side stream on
    the GPU).  One 100-step window with two gloo ranks: exactly one collective, taken after timed step 99, holding all 8 envs of the job."""
    r = _run(["--gpus", "2", "--steps", "100", "--warmup", "0", "--envs-per-gpu", "4", "--repeats", "1"], timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][0])
    g = out["dist"]["periodic_returns_gather"]
    assert g["interval_steps"] == 100 and g["collectives"] == 1
    assert g["last"]["timed_step"] == 99 and g["last"]["envs"] == 8 and abs(g["last"]["mean_return"] - 100 * 3.5) < 1e-3


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_single_rank_line_has_the_contract_keys():
    r = _run(["--steps", "3", "--warmup", "1", "--envs-per-gpu", "4", "--repeats", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "repeats", "first_window", "dist"):
        assert k in out
    assert out["dist"] == {"backend": None, "world_size": 1, "ranks_reporting": 1, "ranks": [0], "per_rank": out["dist"]["per_rank"],
                           "periodic_returns_gather": {"interval_steps": 100, "collectives": 0, "last": None}}
    assert out["n_gpus"] == 1 and out["vs_baseline"] is None and out["scaling"] == "weak"


def test_tree_workload_with_two_ranks_prints_its_line():
    """`bench.py --workload aloha --gpus 2` (the process-group evidence is a collective and was called by rank 0 only - rank 0 then hung):
    the real run_aloha with two gloo ranks and a stub env; one line, both ranks in `dist`, the whole-job value from the slower rank's time."""
    r = _run(["--workload", "aloha", "--gpus", "2", "--steps", "3", "--warmup", "1", "--envs-per-gpu", "8"], timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(line) == 1, r.stdout
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["config"]["global_envs"] == 16 and out["ms_per_step"] >= 30.0 * 0.9
    d = out["dist"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and d["ranks"] == [0, 1] and len(d["per_rank"]) == 2
    assert abs(out["value"] - 16 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    assert abs(out["mean_episode_return"] - 4 * 7.5) < 1e-4
