#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched SO100 HandOver step on MI355X (BASELINE.json metric).

One "step" = one lock-step env.step() of every env on this rank: before_step, 10 physics substeps,
proprioceptive observation gather, reward, discount/termination, auto-reset when an episode ends.

Workloads (BASELINE.json `configs`):
  handover   configs[1] (default): SO100HandOverBanana, 4096 envs per GPU, uniform random actions within
             action_spec, time_limit 10.0 s (500-step episodes), reference reset (placement + 2 s settle,
             prefetched).  `--envs-per-gpu 32768` is the per-GPU share of configs[4].
  pickplace  configs[2]: SO100HandOverBanana, 16384 envs, episodes start from a scripted pre-grasp pool (jaws closing
             on the banana / banana released over the bowl, so101_sim_amd/pregrasp.py), actions = hold pose + N(0, 0.05)
             with the jaw closing; contact-heavy, the reward = 1 branch fires.
  mixed      configs[3]: 16384 x SO100HandOverBanana + 16384 x SO100HandOverPen (two handles, two streams), per-env
             prop mass scale ~ U(0.5, 1.5) on top of the reference's pose randomisation.
Weak scaling: every rank owns --envs-per-gpu envs (global env ids rank*N ...), no data-path collective; episode
returns are all-gathered over RCCL for logging (so101_sim_amd.distributed, outside the timed region).

    python bench.py --gpus 1 --steps 500 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 bench.py --gpus 8 ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # four launch chains need their own hardware queues (so101_sim_amd/__init__.py)

ALGO_BYTES_PER_ENV_STEP = 620      # fused 10-substep step, fp32, per env-step
HBM_SPEC_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3           # MI355X_MICROARCH.md: fp32 vector peak (256 CUs x 4 SIMD x 16 lanes x 2 x 2.4 GHz)
DEFAULT_ENVS = {"handover": 4096, "pickplace": 16384, "mixed": 32768}


def cpu_baseline(seconds_budget: float = 12.0):
    """The fp64 oracle (a port: MuJoCo is not installable here) stepping the handover workload on the host: 1 env
    on ONE core, then one env per core on ALL cores (ctypes releases the GIL, one oracle handle per thread)."""
    import threading
    import numpy as np
    from so101_sim_amd.model import scenes
    from oracle.oracle import Oracle
    raw64, _ = scenes.load_blob("banana", "f64")
    lo = np.array([-np.pi, -3.14158, -3.14158, -3.14158, -3.14158, 0.0])
    hi = np.array([np.pi, 3.14158, 3.14158, 3.14158, 3.14158, 0.08])

    def worker(idx, budget, out):
        o = Oracle(raw64)
        o.env_config(seed=0, env_id=idx, last_step=500)
        rng = np.random.RandomState(1 + idx)
        t0 = time.perf_counter()
        o.env_reset()
        steps = 0
        while time.perf_counter() - t0 < budget:          # episodes follow each other through the auto-reset
            o.env_step(rng.uniform(lo, hi))
            steps += 1
        out[idx] = (steps, time.perf_counter() - t0)

    one = {}
    worker(0, seconds_budget, one)
    ncpu = os.cpu_count() or 1
    many = {}
    th = [threading.Thread(target=worker, args=(i, seconds_budget, many)) for i in range(ncpu)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    total = sum(s for s, _ in many.values())
    return {"value": total / wall, "unit": "env-steps/s", "cores": ncpu, "kind": "port",
            "single_core_value": one[0][0] / one[0][1], "nproc": ncpu,
            "sample": f"handover workload (uniform random actions, 500-step episodes, auto-reset + settle included), fp64 oracle, Newton "
                      f"solver: 1 env on 1 core for {one[0][1]:.1f} s ({one[0][0]} steps), then 1 env per core on {ncpu} cores for {wall:.1f} s "
                      f"({total} steps)"}


def measure_hbm_copy(torch, dev, mib=2048, reps=5):
    """Device-to-device streaming copy, GB/s of read + write traffic: the measured HBM roofline of this box."""
    n = mib * 1024 * 1024 // 4
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        dst.copy_(src)
    ev[1].record()
    torch.cuda.synchronize(dev)
    ms = ev[0].elapsed_time(ev[1]) / reps
    del src, dst
    return 2.0 * n * 4 / (ms * 1e-3) / 1e9


def load_pmc(build_hash):
    """PMC summary (scripts/gpu_pmc.sh -> profiles/pmc_<build hash>.json) of exactly this build, or None."""
    path = os.path.join(ROOT, "profiles", f"pmc_{build_hash}.json")
    if os.path.exists(path):
        try:
            return json.load(open(path))
        except Exception:
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=("handover", "pickplace", "mixed"), default="handover")
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="0 = the workload's BASELINE.json size")
    ap.add_argument("--solver", choices=("newton", "pgs"), default="newton",
                    help="newton = MuJoCo's default, which the reference scene uses (it sets no <option solver>)")
    ap.add_argument("--no-prefetch", action="store_true", help="settle auto-resets inside the step call")
    ap.add_argument("--fused", action="store_true", help="one fused k_step launch per control step instead of the pipeline")
    ap.add_argument("--groups", type=int, default=0, help="env slices of the pipelined step (0 = library default)")
    ap.add_argument("--solver-iterations", type=int, default=0, help="iteration cap; 0 = model default (100)")
    ap.add_argument("--solver-tolerance", type=float, default=-1.0, help="<0 = model default (1e-8)")
    ap.add_argument("--pool-size", type=int, default=4096, help="pickplace: states in the pre-grasp pool")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from so101_sim_amd import distributed as sdist

    rank, local_rank, world = sdist.rank_info()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sdist.init(dev)

    from so101_sim_amd import build as sbuild
    from so101_sim_amd import task_suite
    # roofline denominator first: a few seconds of streaming copies, which also take the GPU out of its idle power state
    # before anything is timed (a fresh box otherwise spends the first timed steps ramping its clocks)
    hbm_measured = measure_hbm_copy(torch, dev, reps=20) if rank == 0 else None
    N = args.envs_per_gpu or DEFAULT_ENVS[args.workload]
    cwd = os.getcwd()
    os.chdir("/tmp")          # calibration offsets OFF (reference looks the JSON up relative to the CWD)
    kw = dict(time_limit=10.0, random_state=0, device=dev, solver_iterations=args.solver_iterations,
              solver_tolerance=args.solver_tolerance, solver=args.solver, prefetch_resets=not args.no_prefetch)
    if args.workload == "mixed":
        half = N // 2
        envs = [task_suite.create_task_env("SO100HandOverBanana", n_envs=half, env_id_base=sdist.shard_base(rank, N), **kw),
                task_suite.create_task_env("SO100HandOverPen", n_envs=N - half, env_id_base=sdist.shard_base(rank, N) + half, **kw)]
    else:
        envs = [task_suite.create_task_env("SO100HandOverBanana", n_envs=N, env_id_base=sdist.shard_base(rank, N), **kw)]
    os.chdir(cwd)
    for env in envs:
        if args.fused:
            env.sim.configure(pipeline=0)
        if args.groups:
            env.sim.configure(groups=args.groups)
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in envs[1:]]
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + rank)
    total = args.warmup + args.steps

    if args.workload == "mixed":
        for env in envs:          # per-env domain randomisation of the props' mass (pose randomisation is the reset's)
            env.set_mass_scale(0.5 + torch.rand(2, env.n_envs, device=dev, generator=gen))
    if args.workload == "pickplace":
        from so101_sim_amd import pregrasp
        pool = pregrasp.build_pickplace_pool(envs[0], pool_size=args.pool_size, seed=rank)
        envs[0].set_reset_pool(*pool)

    spec = envs[0].action_spec()
    lo = torch.tensor(spec.minimum, device=dev)
    hi = torch.tensor(spec.maximum, device=dev)
    if args.workload == "pickplace":
        noise = [0.05 * torch.randn(total, env.n_envs, 6, device=dev, generator=gen) for env in envs]
        hold = [torch.zeros(env.n_envs, 6, device=dev) for env in envs]
    else:
        tapes = [lo + (hi - lo) * torch.rand(total, env.n_envs, 6, device=dev, generator=gen) for env in envs]   # actions resident in HBM

    for env in envs:
        env.reset_all()
    if args.workload == "pickplace":
        hold[0].copy_(envs[0].obs[:, 12:18])

    stats = {k: torch.zeros((), device=dev) for k in ("reward_sum", "ncon", "nefc", "iters", "ncand")}
    stats["samples"] = 0

    def one_step(i, timed, sample=None):
        # `sample`: read the per-env diagnostics after this step.  The first warm-up step does it too (result discarded),
        # so that every torch kernel the sampling needs is loaded before the timed region starts - on a fresh box the
        # lazy load of one kernel costs more than a control step.
        if sample is None:
            sample = timed and (i - args.warmup) % 10 == 0
        for k, env in enumerate(envs):
            with torch.cuda.stream(streams[k]):
                if args.workload == "pickplace":
                    # hold pose of the episode (= commanded pose at FIRST) + small noise, jaw driven 0.3 rad past closed
                    first = (env.step_type == 0).unsqueeze(1)
                    hold[k] = torch.where(first, env.obs[:, 12:18], hold[k])
                    act = hold[k] + noise[k][i]
                    act[:, 5] = hold[k][:, 5] - 0.3 + noise[k][i][:, 5]
                else:
                    act = tapes[k][i]
                env.step_tensor(act)
                if (timed or sample) and args.workload != "handover":      # (the headline workload never reaches reward 1: no extra launches there)
                    stats["reward_sum"] += env.reward.sum()
        if sample:
            for k, env in enumerate(envs):
                with torch.cuda.stream(streams[k]):
                    d = env.diagnostics().float()
                    stats["ncon"] += d[:, 0].sum()
                    stats["nefc"] += d[:, 1].sum()
                    stats["iters"] += d[:, 2].sum()
                    stats["ncand"] += d[:, 3].sum()
            stats["samples"] += sum(e.n_envs for e in envs)

    for i in range(args.warmup):
        one_step(i, False, sample=(i == 0))
    for s in streams[1:]:
        streams[0].wait_stream(s)
    torch.cuda.synchronize(dev)
    for k in stats:
        stats[k] = 0 if k == "samples" else torch.zeros((), device=dev)
    for env in envs:
        env.events(clear=True)
    sdist.barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = time.perf_counter()
    ev[0].record(streams[0])
    for s in streams[1:]:
        s.wait_stream(streams[0])
    for i in range(args.steps):
        one_step(args.warmup + i, True)
    for s in streams[1:]:
        streams[0].wait_stream(s)
    ev[1].record(streams[0])
    torch.cuda.synchronize(dev)
    sdist.barrier()
    elapsed = time.perf_counter() - t0
    # device time of one control step on the stream(s) the kernels are launched on (torch's current stream, which
    # so101_step receives and on which the library's internal slice streams are joined)
    kernel_ms = ev[0].elapsed_time(ev[1]) / args.steps

    # logging-only exchange: episode returns all-gathered over RCCL/xGMI (not in the timed region)
    returns = torch.cat([env.episode_returns() for env in envs])
    all_returns = sdist.all_gather_returns(returns)
    elapsed = sdist.max_over_ranks(elapsed, dev)
    events = {}
    for env in envs:
        for k, v in env.events().items():
            events[k] = events.get(k, 0) + v
    n_local = sum(e.n_envs for e in envs)

    if rank == 0:
        build_hash = sbuild.source_hash()
        value = world * n_local * args.steps / elapsed
        achieved = ALGO_BYTES_PER_ENV_STEP * n_local / (kernel_ms * 1e-3) / 1e9
        pmc = load_pmc(build_hash) if args.workload == "handover" and N == 4096 and not args.fused else None
        env_steps = n_local * args.steps
        names = {"handover": "SO100HandOverBanana, uniform random actions, 500-step episodes with the reference reset (placement + settle, prefetched)",
                 "pickplace": f"SO100HandOverBanana pick-and-place, episodes from a scripted pre-grasp pool of {args.pool_size} states, hold pose + N(0,0.05) actions with the jaw closing",
                 "mixed": "SO100HandOverBanana + SO100HandOverPen halves on two streams, uniform random actions, per-env prop mass scale U(0.5,1.5)"}
        cfg_index = {"handover": 1 if N == 4096 else (4 if N == 32768 else None), "pickplace": 2 if N == 16384 else None, "mixed": 3 if N == 32768 else None}[args.workload]
        out = {
            "metric": "env_steps_per_sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{names[args.workload]}; {n_local} lock-step envs per GPU, proprioceptive obs"
                                   + (f" (BASELINE.json configs[{cfg_index}]" + (" per-GPU share)" if cfg_index == 4 else ")") if cfg_index else ""),
                       "envs_per_gpu": n_local, "global_envs": world * n_local, "substeps_per_step": 10,
                       "solver": args.solver, "reset_prefetch": not args.no_prefetch, "pipeline": not args.fused,
                       "solver_iterations": args.solver_iterations or 100,
                       "solver_tolerance": args.solver_tolerance if args.solver_tolerance >= 0 else 1e-8,
                       "parallelism": f"env-shard x{world}", "build": build_hash},
            "roofline": {"bound": "latency/valu", "achieved": achieved, "peak": hbm_measured, "unit": "GB/s",
                         "frac": achieved / hbm_measured, "traffic": (pmc or {}).get("hbm_bytes_per_step"),
                         "peak_spec": HBM_SPEC_GBS, "frac_of_spec": achieved / HBM_SPEC_GBS,
                         "kernel": "k_step" if args.fused else f"k_order + {args.groups or 4} env slices x (k_pipe_begin + substeps x (k_narrow + k_pipe_solve))",
                         "kernel_ms": kernel_ms, "launches_per_step": 1 if args.fused else 1 + (args.groups or 4) * 21,
                         "compute": None if not pmc else {
                             "valu_tflops_equiv": pmc.get("valu_insts_per_step", 0) * 64 * 2 / (kernel_ms * 1e-3) / 1e12,
                             "peak_tflops": VALU_PEAK_TFLOPS,
                             "frac": pmc.get("valu_insts_per_step", 0) * 64 * 2 / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                             "active_lane_fraction": pmc.get("active_lane_fraction"),
                             "wait_fraction": pmc.get("wait_fraction"), "source": f"profiles/pmc_{build_hash}.json"},
                         "note": "HBM fraction as the metric asks: algorithmic bytes (620 B/env-step x envs) / device time of the step's launch "
                                 "chain / MEASURED device-to-device copy bandwidth of this GPU; the path is bound by dependent-issue latency of "
                                 "wave-level geometry / solver code, not by HBM (DESIGN.md section 7); `compute` = VALU wave-instructions x 64 lanes "
                                 "x 2 flop over the same time against the fp32 vector peak, from the PMC pass of this exact build when one is committed"},
            "diag_mean": {k2: float(stats[k]) / max(stats["samples"], 1) for k, k2 in (("ncon", "contacts"), ("nefc", "constraint_rows"), ("iters", "solver_iterations"), ("ncand", "narrowphase_candidates"))},
            "events_per_env_step": {k: v / env_steps for k, v in events.items()},
            "events": events,
            "mean_reward_per_env_step": float(stats["reward_sum"]) / env_steps,
            "mean_episode_return": float(all_returns.mean().item()),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    sdist.finalize()


if __name__ == "__main__":
    main()
