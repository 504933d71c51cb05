#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched SO100 HandOver step on MI355X (BASELINE.json metric).

One "step" = one lock-step env.step() of every env on this rank: before_step, 10 physics substeps,
proprioceptive observation gather, reward, discount/termination, auto-reset (with the reference's
settle) when an episode ends.  Workload = BASELINE.json configs[1]: SO100HandOverBanana, 4096 envs
per GPU, proprioceptive obs only, uniform random actions within action_spec, time_limit 10.0 s
(500-step episodes), calibration offsets off.  Weak scaling: every rank owns 4096 envs (global env ids
rank*4096 ...), no data-path collective; episode returns are all-gathered over RCCL for logging.

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

ALGO_BYTES_PER_ENV_STEP = 620      # fused 10-substep step, fp32, per env-step
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(seconds_budget: float = 20.0):
    """The fp64 oracle (a port: MuJoCo is not installable here) stepping the same workload on ONE host
    core: 1 env, reset + random-action steps until the time budget is used."""
    import numpy as np
    from so101_sim_amd.model import scenes
    from oracle.oracle import Oracle
    raw64, _ = scenes.load_blob("banana", "f64")
    o = Oracle(raw64)
    o.env_config(seed=0, env_id=0, last_step=500)
    rng = np.random.RandomState(1)
    lo = np.array([-np.pi, -3.14158, -3.14158, -3.14158, -3.14158, 0.0])
    hi = np.array([np.pi, 3.14158, 3.14158, 3.14158, 3.14158, 0.08])
    t0 = time.perf_counter()
    o.env_reset()
    steps = 0
    while time.perf_counter() - t0 < seconds_budget:          # episodes follow each other through the auto-reset
        o.env_step(rng.uniform(lo, hi))
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"1 env, {steps} env.step calls with uniform random actions in {dt:.1f} s (500-step episodes, auto-reset + settle "
                      "included), fp64 oracle, Newton solver"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--solver", choices=("newton", "pgs"), default="newton",
                    help="newton = MuJoCo's default, which the reference scene uses (it sets no <option solver>)")
    ap.add_argument("--no-prefetch", action="store_true", help="settle auto-resets inside the step call")
    ap.add_argument("--fused", action="store_true", help="one fused k_step launch per control step instead of the pipeline")
    ap.add_argument("--groups", type=int, default=0, help="env slices of the pipelined step (0 = library default)")
    ap.add_argument("--solver-iterations", type=int, default=0, help="iteration cap; 0 = model default (100)")
    ap.add_argument("--solver-tolerance", type=float, default=-1.0, help="<0 = model default (1e-8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from so101_sim_amd import task_suite
    N = args.envs_per_gpu
    cwd = os.getcwd()
    os.chdir("/tmp")          # calibration offsets OFF (reference looks the JSON up relative to the CWD)
    env = task_suite.create_task_env("SO100HandOverBanana", time_limit=10.0, random_state=0, n_envs=N,
                                     device=dev, env_id_base=rank * N, solver_iterations=args.solver_iterations,
                                     solver_tolerance=args.solver_tolerance, solver=args.solver,
                                     prefetch_resets=not args.no_prefetch)
    os.chdir(cwd)
    if args.fused:
        env.sim.configure(pipeline=0)
    if args.groups:
        env.sim.configure(groups=args.groups)
    spec = env.action_spec()
    lo = torch.tensor(spec.minimum, device=dev)
    hi = torch.tensor(spec.maximum, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + rank)
    total = args.warmup + args.steps
    tape = lo + (hi - lo) * torch.rand(total, N, 6, device=dev, generator=gen)   # actions resident in HBM

    env.reset_all()
    for i in range(args.warmup):
        env.step_tensor(tape[i])
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        env.step_tensor(tape[args.warmup + i])
    ev[1].record()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # device time of one control step on the stream the kernels are launched on (torch's current stream, which
    # so101_step receives and on which the internal slice streams are joined): k_order + per env slice
    # 1 + 2*substeps launches with the pipelined step, one k_step launch with --fused
    kernel_ms = ev[0].elapsed_time(ev[1]) / args.steps

    # logging-only exchange: episode returns all-gathered over RCCL/xGMI (not in the timed region)
    returns = env.episode_returns()
    if distributed:
        gathered = [torch.empty_like(returns) for _ in range(world)]
        dist.all_gather(gathered, returns)
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        mean_return = float(torch.cat(gathered).mean().item())
    else:
        mean_return = float(returns.mean().item())
    diag = env.diagnostics().float().mean(0).tolist()

    if rank == 0:
        value = world * N * args.steps / elapsed
        achieved = ALGO_BYTES_PER_ENV_STEP * N / (kernel_ms * 1e-3) / 1e9
        traffic = None
        # HBM bytes per control step from the committed PMC passes (FETCH_SIZE x2 per the gfx950 note in
        # MI355X_MICROARCH.md + WRITE_SIZE, summed over the launches of one step); not measured live
        prof = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(prof) and not args.fused:
            try:
                traffic = json.load(open(prof)).get("hbm_bytes_per_step")
            except Exception:
                traffic = None
        out = {
            "metric": "env_steps_per_sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"SO100HandOverBanana, {N} lock-step envs per GPU, proprioceptive obs, uniform random actions, 500-step episodes with auto-reset+settle" + (" (BASELINE.json configs[1])" if N == 4096 else ""),
                       "envs_per_gpu": N, "global_envs": world * N, "substeps_per_step": 10,
                       "solver": args.solver, "reset_prefetch": not args.no_prefetch, "pipeline": not args.fused, "solver_iterations": args.solver_iterations or 100,
                       "solver_tolerance": args.solver_tolerance if args.solver_tolerance >= 0 else 1e-8,
                       "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_step" if args.fused else "k_order + 3 env slices x (k_pipe_begin + substeps x (k_narrow + k_pipe_solve))",
                         "kernel_ms": kernel_ms, "launches_per_step": 1 if args.fused else 1 + (args.groups or 3) * 21,
                         "note": "per control step: algorithmic bytes = 620 B/env-step x envs, time = device time of the step's "
                                 "launch chain; the path is latency/VALU-bound, not HBM-bound (DESIGN.md section 6)"},
            "diag_mean": {"ncon": diag[0], "nefc": diag[1], "solver_iter": diag[2], "broadphase_candidates": diag[3]},
            "mean_episode_return": mean_return,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
