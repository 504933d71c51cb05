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
  aloha      (not a BASELINE.json config) HandOverBanana of the ALOHA bimanual robot on the general-tree engine,
             4096 envs per GPU, random joint targets around the home pose.
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
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # four launch chains need their own hardware queues (so101_sim_amd/__init__.py)

ALGO_BYTES_PER_ENV_STEP = 620      # fused 10-substep step, fp32, per env-step
HBM_SPEC_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3           # MI355X_MICROARCH.md: fp32 vector peak (256 CUs x 4 SIMDs x 32 lanes x 2 flop x 2.4 GHz)
VALU_SUSTAINED_TINST = 0.81        # T wave64 VALU instructions/s the device sustains on plain v_fma_f32 with >= 2 wavefronts per SIMD, measured
                                   # (scripts/microbench/valu_issue.hip, profiles/r06_valu_issue.txt: 103 TFLOP/s; the clock sags to 1.5-2.0 GHz under that load)
DEFAULT_ENVS = {"handover": 4096, "pickplace": 16384, "mixed": 32768}


def usable_cores() -> int:
    """Cores this process may actually use: the affinity mask, cut by the cgroup CPU quota (a container on a 256-thread host
    with a quota of 16 CPUs reports os.cpu_count() = 256, and 256 busy threads then share 16 CPUs)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(envs_per_worker: int = 64, steps: int = 6, reps: int = 5):
    """The fp64 oracle (a port: MuJoCo is not installable here) on the host cores: C++ threads inside the oracle
    library (oracle/so101_oracle.cpp orc_bench_rollout), `envs_per_worker` envs per thread, 1 thread and then one thread
    per core.  Times placement + settle of every env, then `reps` repetitions of `steps` control steps of every env
    under uniform random actions, for two solver settings: the GPU's (Newton, 100 iterations, tolerance 1e-8) and a
    fixed-iteration one (Newton, 4 iterations, tolerance 0: above the GPU's measured mean of 2.5-3.2).  `value` =
    env-steps/s on all cores at the GPU's setting with the reset amortised over 500-step episodes."""
    import ctypes as C
    import numpy as np
    from so101_sim_amd.model import scenes
    from oracle import oracle as orc
    raw64, _ = scenes.load_blob("banana", "f64")
    L = orc.lib()
    L.orc_bench_rollout.restype = C.c_longlong
    L.orc_bench_rollout.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                    C.POINTER(C.c_double), C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    settings = [("newton_100it_tol1e-8", 100, 1e-8), ("newton_fixed_4it", 4, 0.0)]
    its = (C.c_int * 2)(*[x[1] for x in settings])
    tol = (C.c_double * 2)(*[x[2] for x in settings])
    ncpu = usable_cores()

    def leg(threads):
        rs = C.c_double(0.0)
        rep = (C.c_double * (2 * reps))()
        n = L.orc_bench_rollout(raw64, len(raw64), threads, envs_per_worker, steps, reps, 2, its, tol, 0, C.byref(rs), rep)
        if n <= 0:
            raise RuntimeError("orc_bench_rollout failed")
        out = {"threads": threads, "envs": threads * envs_per_worker, "reset_seconds": rs.value,
               "reset_seconds_per_env_per_core": rs.value / envs_per_worker}
        for k, (name, _, _) in enumerate(settings):
            r = np.array([n / rep[k * reps + j] for j in range(reps)])
            # one 500-step episode of one env costs 500 steps + one reset on its core
            step_s = threads / r.mean()                      # seconds of one core per env-step
            amort = threads * 500.0 / (500.0 * step_s + rs.value / envs_per_worker)
            out[name] = {"env_steps_per_s_mean": float(r.mean()), "env_steps_per_s_std": float(r.std()), "reps": reps,
                         "with_reset_amortised_over_500_steps": float(amort)}
        return out

    one, many = leg(1), leg(ncpu)
    key = settings[0][0]
    eff = many[key]["env_steps_per_s_mean"] / (ncpu * one[key]["env_steps_per_s_mean"])
    return {"value": many[key]["with_reset_amortised_over_500_steps"], "unit": "env-steps/s", "cores": ncpu, "kind": "port",
            "host_logical_cpus": os.cpu_count(),
            "label": "CPU restatement baseline (MuJoCo unavailable)",
            "single_core_value": one[key]["with_reset_amortised_over_500_steps"], "parallel_efficiency": eff,
            "one_thread": one, "all_threads": many,
            "sample": f"handover workload, fp64 oracle, C++ threads, {envs_per_worker} envs per thread: placement + settle of every env "
                      f"(timed on its own), then {reps} repetitions of {steps} control steps of every env under uniform random actions "
                      f"per solver setting (first steps of the episode: ~9 contacts per env); 1 thread, then {ncpu} threads; "
                      f"value = all threads, GPU solver setting, reset amortised over 500-step episodes; mean +- std over the repetitions inside"}


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


def spawn_ranks(n: int) -> int:
    """`bench.py --gpus N` launched plainly: start N fresh rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, one GPU each, RCCL or gloo rendezvous on 127.0.0.1), wait for them and relay rank 0's JSON line.  This
    parent never imports torch and never touches HIP; nothing re-execs."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True if r == 0 else None))
    # rank 0's line is read by a thread while ALL children are polled: a rank that dies before or during the rendezvous would
    # otherwise leave rank 0 (and this parent, blocked in communicate()) waiting for the process group's timeout
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rcs = [None] * n
    while any(c is None for c in rcs):
        for r, q in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = q.poll()
        failed = [r for r, c in enumerate(rcs) if c not in (None, 0)]
        if failed:
            for r, q in enumerate(procs):
                if rcs[r] is None:
                    q.terminate()
            for r, q in enumerate(procs):
                if rcs[r] is None:
                    try:
                        rcs[r] = q.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        q.kill()
                        rcs[r] = q.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write("".join(c or "" for c in chunks))
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        sys.stderr.write(f"bench.py: rank(s) failed: {bad}\n")
        return 1
    return 0


def resolve_factory(spec: str):
    """`module:callable` -> callable(name, n_envs, env_id_base, device, **kw) (tests stub the env with it)."""
    import importlib
    mod, _, fn = spec.partition(":")
    return getattr(importlib.import_module(mod), fn)


def run_aloha(args, torch, sdist, dev, rank, world, hbm_measured):
    """`--workload aloha`: HandOverBanana on the general-tree engine - the ALOHA bimanual robot, nq 30 / nv 28 /
    nu 14 -, uniform random joint targets around the home pose, resets inside the step calls.  Same contract as the other
    workloads: W untimed steps, K timed steps between barrier + synchronize, max over ranks, one JSON line."""
    import numpy as np
    from so101_sim_amd import build as sbuild, task_suite
    from so101_sim_amd.model import scenes
    dining = args.workload == "dining"
    N = args.envs_per_gpu or (1024 if dining else 4096)
    task_name = "DiningPlaceBananaInBowl" if dining else "HandOverBanana"
    on_gpu = args.device == "cuda"
    sync = (lambda: torch.cuda.synchronize(dev)) if on_gpu else (lambda: None)
    if args.env_factory:            # tests: a stub env (tests/bench_stub.py) in place of the GPU env, gloo ranks on CPU
        env = resolve_factory(args.env_factory)(task_name, N, sdist.shard_base(rank, N), dev, workload=args.workload)
    else:
        env = task_suite.create_task_env(task_name, time_limit=10.0, random_state=0, n_envs=N, device=dev, narrowphase=args.narrowphase,
                                         env_id_base=sdist.shard_base(rank, N), solver_iterations=args.solver_iterations, solver_tolerance=args.solver_tolerance)
    if args.settled_store:          # placement + settle of the episodes the run will start, done once before anything is timed
        t_store = time.perf_counter(); env.compute_settled(2 + (args.warmup + args.steps) // 500); t_store = time.perf_counter() - t_store
    env.reset()
    gen = torch.Generator(device=dev); gen.manual_seed(1 + rank)
    spec = env.action_spec()
    lo, hi = torch.tensor(spec.minimum, device=dev), torch.tensor(spec.maximum, device=dev)
    home = torch.tensor(np.concatenate([scenes.ALOHA_HOME_CTRL] * 2), dtype=torch.float32, device=dev)
    total = args.warmup + args.steps
    tape = torch.clamp(home + 0.5 * (torch.rand(total, N, 14, device=dev, generator=gen) - 0.5), lo, hi)     # actions resident in HBM
    for k in range(args.warmup):
        env.step_tensor(tape[k])
    sdist.barrier(); sync()
    t0 = time.perf_counter()
    reward_sum = torch.zeros((), device=dev)
    for k in range(args.steps):
        _, r, _, _ = env.step_tensor(tape[args.warmup + k])
        reward_sum += r.sum()
    sdist.barrier(); sync()
    elapsed = sdist.max_over_ranks(time.perf_counter() - t0, dev)
    d = env.diagnostics().cpu().numpy()
    all_returns = sdist.all_gather_returns(env.episode_returns())
    dist_info = sdist.evidence(1e3 * elapsed / args.steps, dev)      # a collective: every rank calls it
    plan = env.launch_plan()
    if rank == 0:
        # counters of exactly this build's control step (the launch chain) at 4096 ALOHA envs, when a PMC pass of it is committed (scripts/gpu_pmc_tree.sh)
        tree_hash = sbuild.source_hash(mpr=args.narrowphase == "mpr")
        tree_pmc = None
        pmc_path = os.path.join(ROOT, "profiles", f"pmc_tree_{tree_hash}.json")
        if not dining and N == 4096 and os.path.exists(pmc_path):
            with open(pmc_path) as fh:
                tree_pmc = json.load(fh)
        nq, nv = env.sim.nq, env.sim.nv
        algo = 4 * (14 + 2 * (nq + nv) + 2 * nv + 74 + 2 * (14 + 16) + 2 + 4) + 1      # action, state r+w, warm start r+w, obs, delay lines r+w, reward / discount, counters, step type
        achieved = algo * N / (elapsed / args.steps) / 1e9
        print(json.dumps({
            "metric": "env_steps_per_sec", "value": world * N * args.steps / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{task_name} ({'Dining scene: ALOHA + six free props, task_suite.py:54' if dining else 'ALOHA bimanual, task_suite.py:63'}; nq {nq} / nv {nv} / nu 14), "
                                    f"uniform random joint targets around the home pose, 500-step episodes with the reference reset inside the step calls; {N} lock-step "
                                    f"envs per GPU, proprioceptive obs (not a BASELINE.json config: SURVEY 8f-{'4' if dining else '1'})"),
                       "envs_per_gpu": N, "global_envs": world * N, "substeps_per_step": 10, "solver": "newton", "narrowphase": args.narrowphase,
                       "engine": f"general tree, {env.sim.build}-dof build (csrc/so101_tree.hpp)",
                       "resets": (f"settled-state store computed before the timed region ({t_store:.1f} s for {2 + (args.warmup + args.steps) // 500} episodes per env)"
                                  if args.settled_store else "placement + settle inside the step calls"),
                       "parallelism": f"env-shard x{world}", "build": sbuild.source_hash(mpr=args.narrowphase == "mpr")},
            "roofline": {"bound": "latency/valu", "achieved": achieved, "peak": hbm_measured, "unit": "GB/s", "frac": achieved / hbm_measured if hbm_measured else None,
                         "traffic": (tree_pmc or {}).get("hbm_bytes_per_step"), "peak_spec": HBM_SPEC_GBS, "frac_of_spec": achieved / HBM_SPEC_GBS,
                         "kernel": f"{plan['path']}: {plan.get('kernels', 'k_tree_step')}; {plan['slices']} env slice(s) on their own streams",
                         "kernel_ms": 1e3 * elapsed / args.steps, "launches_per_step": plan["kernel_launches"], "memsets_per_step": plan["memsets"],
                         "compute": None if not tree_pmc else {
                             "valu_tflops_equiv": tree_pmc["valu_insts_per_step"] * 64 * 2 / (elapsed / args.steps) / 1e12, "peak_tflops": VALU_PEAK_TFLOPS,
                             "frac": tree_pmc["valu_insts_per_step"] * 64 * 2 / (elapsed / args.steps) / 1e12 / VALU_PEAK_TFLOPS,
                             "valu_insts_per_env_substep": tree_pmc["valu_insts_per_step"] / (N * 10.0), "active_lane_fraction": tree_pmc.get("active_lane_fraction"),
                             "wait_fraction": tree_pmc.get("wait_fraction"), "source": f"profiles/pmc_tree_{tree_hash}.json"},
                         "note": f"algorithmic bytes {algo} B per env-step; the step is bound by its instruction count and the per-CU LDS / L1 traffic of the solver "
                                 "(DESIGN.md section 8), not by HBM"},
            "dist": dist_info,
            "windows": "first window only (the SO100 workloads report the mean of --repeats windows)",
            "diag_mean": {"contacts": float(d[:, 0].mean()), "constraint_rows": float(d[:, 1].mean()), "solver_iterations": float(d[:, 2].mean()),
                          "narrowphase_candidates": float(d[:, 3].mean())},
            "flagged_envs_last_step": int((d[:, 4] != 0).sum()),
            "mean_reward_per_env_step": float(reward_sum) / (N * args.steps), "mean_episode_return": float(all_returns.mean().item())}))
    env.close()
    sdist.finalize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=("handover", "pickplace", "mixed", "aloha", "dining"), default="handover")
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="0 = the workload's BASELINE.json size")
    ap.add_argument("--solver", choices=("newton", "pgs"), default="newton",
                    help="newton = MuJoCo's default, which the reference scene uses (it sets no <option solver>)")
    ap.add_argument("--narrowphase", choices=("mpr", "epa"), default="epa", help="epa (default): minimum translation, what mujoco >= 3.3 reports; mpr: the -DSO101_MPR build of the library (MPR's portal depth, built on demand)")
    ap.add_argument("--no-prefetch", action="store_true", help="settle auto-resets inside the step call")
    ap.add_argument("--fused", action="store_true", help="one fused k_step launch per control step instead of the pipeline")
    ap.add_argument("--pipeline", type=int, default=-1, help="step path: 3 merged launches, 2 per-env chained, 1 launch chains, 0 fused (-1 = library default)")
    ap.add_argument("--chain-waves", type=int, default=0, help="pipeline 2: persistent wavefronts (0 = library default)")
    ap.add_argument("--no-graph", action="store_true", help="plain launches instead of HIP-graph replay")
    ap.add_argument("--groups", type=int, default=0, help="env slices of the pipelined step (0 = library default)")
    ap.add_argument("--solver-iterations", type=int, default=0, help="iteration cap; 0 = model default (100)")
    ap.add_argument("--solver-tolerance", type=float, default=-1.0, help="<0 = model default (1e-8)")
    ap.add_argument("--pool-size", type=int, default=4096, help="pickplace: states in the pre-grasp pool")
    ap.add_argument("--settled-store", action="store_true", help="aloha: precompute the settled reset states of the run's episodes (outside the timed region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=5, help="timed windows of --steps steps; `value` is their mean, all are in `repeats`")
    ap.add_argument("--device", choices=("cuda", "cpu"), default="cuda", help="cpu: tests only (gloo, needs --env-factory)")
    ap.add_argument("--env-factory", default="", help="module:callable replacing task_suite.create_task_env (tests)")
    args = ap.parse_args()

    if args.pipeline >= 2 and "SO101_HIP_LIB" not in os.environ:
        # the experimental step paths (per-env chaining, merged launches) are not in the default library: build libso101_hip_exp.so and load it
        from so101_sim_amd import build as _b
        os.environ["SO101_HIP_LIB"] = _b.build(exp=True, mpr=args.narrowphase == "mpr")
    # A plain `bench.py --gpus N` (no torchrun environment) launches its own N ranks before anything touches the GPU
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    from so101_sim_amd import distributed as sdist

    rank, local_rank, world = sdist.rank_info()
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}\n")
        sys.exit(2)
    on_gpu = args.device == "cuda"
    if on_gpu:
        if torch.cuda.device_count() <= local_rank:
            sys.stderr.write(f"bench.py: rank {rank} needs GPU {local_rank} but only {torch.cuda.device_count()} device(s) are visible\n")
            sys.exit(3)
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")
    sdist.init(dev)
    assert (not torch.distributed.is_initialized() and world == 1) or torch.distributed.get_world_size() == args.gpus

    from so101_sim_amd import build as sbuild
    # roofline denominator first: a few seconds of streaming copies, which also take the GPU out of its idle power state
    # before anything is timed (a fresh box otherwise spends the first timed steps ramping its clocks)
    hbm_measured = measure_hbm_copy(torch, dev, reps=20) if (rank == 0 and on_gpu) else None
    if args.workload in ("aloha", "dining"):
        return run_aloha(args, torch, sdist, dev, rank, world, hbm_measured)
    N = args.envs_per_gpu or DEFAULT_ENVS[args.workload]
    cwd = os.getcwd()
    os.chdir("/tmp")          # calibration offsets OFF (reference looks the JSON up relative to the CWD)
    kw = dict(time_limit=10.0, random_state=0, device=dev, solver_iterations=args.solver_iterations,
              solver_tolerance=args.solver_tolerance, solver=args.solver, prefetch_resets=not args.no_prefetch)
    if args.narrowphase != "epa":
        kw["narrowphase"] = args.narrowphase
    if args.env_factory:
        make = resolve_factory(args.env_factory)
    else:
        from so101_sim_amd import task_suite
        make = task_suite.create_task_env
    if args.workload == "mixed":
        half = N // 2
        envs = [make("SO100HandOverBanana", n_envs=half, env_id_base=sdist.shard_base(rank, N), **kw),
                make("SO100HandOverPen", n_envs=N - half, env_id_base=sdist.shard_base(rank, N) + half, **kw)]
    else:
        envs = [make("SO100HandOverBanana", n_envs=N, env_id_base=sdist.shard_base(rank, N), **kw)]
    os.chdir(cwd)
    for env in envs:
        if args.fused:
            env.sim.configure(pipeline=0)
        elif args.pipeline >= 0:
            env.sim.configure(pipeline=args.pipeline)
        if args.chain_waves:
            env.sim.configure(chain_waves=args.chain_waves)
        if args.no_graph:
            env.sim.configure(use_graph=0)
        if args.groups:
            env.sim.configure(groups=args.groups)

    # Every env steps on its own non-default stream (the library captures its launch sequence into a HIP graph, which the
    # legacy null stream cannot do); resets, tapes and statistics of an env are produced and consumed on that stream only.
    class _NoStream:
        def wait_stream(self, other): pass
        def synchronize(self): pass
    if on_gpu:
        streams = [torch.cuda.Stream(dev) for _ in envs]
        on_stream = torch.cuda.stream
        sync = lambda: torch.cuda.synchronize(dev)
        class Tick:
            def __init__(self, stream):
                self.ev = torch.cuda.Event(enable_timing=True); self.ev.record(stream)
            def ms_until(self, other):
                return self.ev.elapsed_time(other.ev)
    else:
        import contextlib
        streams = [_NoStream() for _ in envs]
        on_stream = lambda s: contextlib.nullcontext()
        sync = lambda: None
        class Tick:
            def __init__(self, stream):
                self.t = time.perf_counter()
            def ms_until(self, other):
                return 1e3 * (other.t - self.t)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + rank)
    # every timed window gets its own slice of the action tape (memory permitting: 24 B per env-step)
    n_windows = max(args.repeats, 1)
    if (args.warmup + n_windows * args.steps) * N * 24 > 6e9:
        n_windows = 1
    total = args.warmup + n_windows * args.steps

    spec = envs[0].action_spec()
    lo = torch.tensor(spec.minimum, device=dev)
    hi = torch.tensor(spec.maximum, device=dev)
    tapes, noise, hold = [None] * len(envs), [None] * len(envs), [None] * len(envs)
    stats = [{k: torch.zeros((), device=dev) for k in ("reward_sum", "ncon", "nefc", "iters", "ncand")} for _ in envs]
    samples = [0]
    for k, env in enumerate(envs):
        with on_stream(streams[k]):
            if args.workload == "mixed":      # per-env domain randomisation of the props' mass (pose randomisation is the reset's)
                env.set_mass_scale(0.5 + torch.rand(2, env.n_envs, device=dev, generator=gen))
            if args.workload == "pickplace":
                from so101_sim_amd import pregrasp
                # (built once, on rank 0, and broadcast: the pool is the same on every rank, only the draws from it differ per env)
                env.set_reset_pool(*sdist.build_on_rank0(lambda: pregrasp.build_pickplace_pool(env, pool_size=args.pool_size, seed=0)))
                noise[k] = 0.05 * torch.randn(total, env.n_envs, 6, device=dev, generator=gen)
            else:
                tapes[k] = lo + (hi - lo) * torch.rand(total, env.n_envs, 6, device=dev, generator=gen)   # actions resident in HBM
            env.reset_all()
            if args.workload == "pickplace":
                hold[k] = env.obs[:, 12:18].clone()
    sync()

    # the path's one exchange step (SURVEY 8e, configs[4]): every 100 control steps the episode returns of every rank are all-gathered for
    # logging, on a side stream behind a snapshot - inside the timed region whenever a window is that long (the driver's 20-step windows
    # never reach it; `--steps 500` does five)
    returns_log = sdist.PeriodicReturnsGather(interval=100, device=dev)
    steps_done = [0]

    def one_step(i, timed, sample=None):
        # `i` indexes the action tape (warm-up part, then one K-step slice per timed window).  `sample`: read the per-env diagnostics after this step.  The first warm-up
        # step does it too (result discarded), so that every torch kernel the sampling needs is loaded before the timed
        # region starts - on a fresh box the lazy load of one kernel costs more than a control step.
        if sample is None:
            sample = timed and (i - args.warmup) % 10 == 0 and i < args.warmup + args.steps
        for k, env in enumerate(envs):
            with on_stream(streams[k]):
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
                    stats[k]["reward_sum"] += env.reward.sum()
                if sample:
                    d = env.diagnostics().float()
                    stats[k]["ncon"] += d[:, 0].sum()
                    stats[k]["nefc"] += d[:, 1].sum()
                    stats[k]["iters"] += d[:, 2].sum()
                    stats[k]["ncand"] += d[:, 3].sum()
        if sample:
            samples[0] += sum(e.n_envs for e in envs)
        if timed:
            with on_stream(streams[0]):
                if len(envs) > 1:
                    streams[0].wait_stream(streams[1])
                returns_log.maybe(steps_done[0], lambda: torch.cat([env.episode_returns() for env in envs]))
            steps_done[0] += 1

    for i in range(args.warmup):
        one_step(i, False, sample=(i == 0))
    sync()
    for st in stats:
        for k in st:
            st[k].zero_()
    samples[0] = 0
    for env in envs:
        env.events(clear=True)
        if hasattr(env, "sim") and hasattr(env.sim, "chain_stats"):
            env.sim.chain_stats(clear=True)

    drain = [0.0]

    def timed_window(segment=0, window=0):
        """EXACTLY --steps steps bracketed by barrier + synchronize on both sides.  Returns (host seconds, device ms per
        step on the launch streams, [device ms of each `segment`-step slice])."""
        sync()
        sdist.barrier()
        t0 = time.perf_counter()
        ticks = [[Tick(s)] for s in streams]
        base = args.warmup + (window % n_windows) * args.steps
        for i in range(args.steps):
            one_step(base + i, True)
            if segment and (i + 1) % segment == 0 and i + 1 < args.steps:
                for k, s in enumerate(streams):
                    ticks[k].append(Tick(s))
        for k, s in enumerate(streams):
            ticks[k].append(Tick(s))
        # the K steps are complete when their launch streams are (what a caller that joins ITS stream waits for) ...
        for s in streams:
            s.synchronize()
        joined = time.perf_counter() - t0
        # ... and the contract's device-wide synchronize also waits for work the steps started for FUTURE episodes: the reset prefetch
        # that the last mass reset of a long window kicked off on the library's low-priority stream (up to ~2 s on 256 wavefronts)
        sync()
        full = time.perf_counter() - t0
        # The clock of the window stops at the device-wide synchronize - unless that one outlasts the steps by more than 1 %: then it waited
        # for the background prefetch, not for anything the K steps produced, and the window's time is the join of the launch streams (both
        # figures go into the line: `sustained.prefetch_drain_s`, `closing_sync`).  The driver's 20-step windows never see the difference.
        own = joined if full - joined > 0.01 * joined else full      # this rank's own time, before it waits for the others
        sdist.barrier()
        host = own + (time.perf_counter() - t0 - full)
        drain[0] = full - joined
        dev_ms = max(t[0].ms_until(t[-1]) for t in ticks) / args.steps
        segs = [max(t[j].ms_until(t[j + 1]) for t in ticks) for j in range(len(ticks[0]) - 1)] if segment else []
        return host, dev_ms, segs, own, joined

    elapsed_local, kernel_ms, segments, own_local, joined_local = timed_window(segment=100 if args.steps >= 500 else 0)
    elapsed = sdist.max_over_ranks(elapsed_local, dev)
    joined = sdist.max_over_ranks(joined_local, dev)
    events_first = {}
    for env in envs:
        for k, v in env.events().items():
            events_first[k] = events_first.get(k, 0) + v
    stats_first = {k: sum(float(st[k]) for st in stats) for k in stats[0]}
    stats_first["samples"] = samples[0]
    # Further windows of the same length for the spread of the measurement.  They run back to back WITHOUT draining the GPU in
    # between (device events on the launch streams at the window boundaries, one synchronize at the end): windows separated by
    # a host synchronisation measure the clock governor, not the kernels - with the GPU idling for a moment every 20 steps the
    # same steps ran at 630 k, 480 k, 420 k, 390 k env-steps/s against a steady 740-750 k back to back (scripts/gpu_windows.py).
    rep_elapsed = [elapsed]
    n_more = max(args.repeats, 1) - 1
    if n_more:
        sdist.barrier()
        marks = [[Tick(s)] for s in streams]
        for r in range(1, n_more + 1):
            base = args.warmup + (r % n_windows) * args.steps
            for i in range(args.steps):
                one_step(base + i, True, sample=False)
            for k, s in enumerate(streams):
                marks[k].append(Tick(s))
        sync()
        sdist.barrier()
        for r in range(n_more):
            rep_elapsed.append(sdist.max_over_ranks(1e-3 * max(m[r].ms_until(m[r + 1]) for m in marks), dev))

    # what the process group itself reports (backend, world size, which ranks answered) + every rank's own ms per step of the first window
    dist_info = sdist.evidence(1e3 * own_local / args.steps, dev)
    logged = returns_log.latest()
    dist_info["periodic_returns_gather"] = {"interval_steps": returns_log.interval, "collectives": returns_log.count,
                                            "last": None if logged is None else {"timed_step": logged[0], "envs": int(logged[1].numel()), "mean_return": float(logged[1].float().mean())}}
    # logging-only exchange: episode returns all-gathered over RCCL/xGMI (not in the timed region)
    returns = torch.cat([env.episode_returns() for env in envs])
    all_returns = sdist.all_gather_returns(returns)
    events = events_first
    n_local = sum(e.n_envs for e in envs)
    stats = stats_first

    path = 0 if args.fused else (args.pipeline if args.pipeline >= 0 else 1)
    if rank == 0:
        build_hash = sbuild.source_hash(mpr=args.narrowphase == "mpr", exp=args.pipeline >= 2)      # (the library that ran: pipeline 2 / 3 live in the exp build)
        # `value` = the MEAN over all timed windows (each EXACTLY --steps steps; the first one host-clocked between barrier +
        # synchronize, the others back to back behind it, cut with device events), not the first - and usually best - window alone
        rep_values = [world * n_local * args.steps / e for e in rep_elapsed]
        first_window = rep_values[0]
        elapsed_mean = sum(rep_elapsed) / len(rep_elapsed)
        value = world * n_local * args.steps / elapsed_mean
        achieved = ALGO_BYTES_PER_ENV_STEP * n_local / (kernel_ms * 1e-3) / 1e9
        pmc = load_pmc(build_hash) if args.workload == "handover" and N == 4096 and path == 1 else None
        env_steps = n_local * args.steps
        names = {"handover": "SO100HandOverBanana, uniform random actions, 500-step episodes with the reference reset (placement + settle, prefetched)",
                 "pickplace": f"SO100HandOverBanana pick-and-place, episodes from a scripted pre-grasp pool of {args.pool_size} states, hold pose + N(0,0.05) actions with the jaw closing",
                 "mixed": "SO100HandOverBanana + SO100HandOverPen halves on two streams, uniform random actions, per-env prop mass scale U(0.5,1.5)"}
        cfg_index = {"handover": 1 if N == 4096 else (4 if N == 32768 else None), "pickplace": 2 if N == 16384 else None, "mixed": 3 if N == 32768 else None}[args.workload]
        out = {
            "metric": "env_steps_per_sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed_mean / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "first_window": {"value": first_window, "ms_per_step": 1e3 * elapsed / args.steps,
                             "closing_sync": "device-wide synchronize" if drain[0] <= 0.01 * joined else
                                             f"launch streams joined; the device-wide synchronize returned {drain[0]:.2f} s later (background reset prefetch of future episodes)"},
            "dist": dist_info,
            "config": {"workload": f"{names[args.workload]}; {n_local} lock-step envs per GPU, proprioceptive obs"
                                   + (f" (BASELINE.json configs[{cfg_index}]" + (" per-GPU share)" if cfg_index == 4 else ")") if cfg_index else ""),
                       "envs_per_gpu": n_local, "global_envs": world * n_local, "substeps_per_step": 10,
                       "solver": args.solver, "narrowphase": args.narrowphase, "reset_prefetch": not args.no_prefetch, "pipeline": path,
                       "solver_iterations": args.solver_iterations or 100,
                       "solver_tolerance": args.solver_tolerance if args.solver_tolerance >= 0 else 1e-8,
                       "parallelism": f"env-shard x{world}", "build": build_hash},
            "roofline": {"bound": "latency/valu", "achieved": achieved, "peak": hbm_measured, "unit": "GB/s",
                         "frac": achieved / hbm_measured if hbm_measured else None, "traffic": (pmc or {}).get("hbm_bytes_per_step"),
                         "peak_spec": HBM_SPEC_GBS, "frac_of_spec": achieved / HBM_SPEC_GBS,
                         "kernel": {0: "k_step", 1: f"k_order + {args.groups or 4} env slices x (k_pipe_begin + substeps x (k_narrow + k_pipe_solve))",
                                    2: "k_order + k_pipe_begin + k_chain (persistent: narrowphase chunks and per-env solve items from device-side queues)",
                                    3: f"k_order + {args.groups or 4} env slices x (k_pipe_begin + (1 + substeps) x k_pipe_merged: solve of substep s, then narrowphase chunks of s + 1)"}[path],
                         "kernel_ms": kernel_ms, "launches_per_step": {0: 1, 1: 1 + (args.groups or 4) * 21, 2: 3, 3: 1 + (args.groups or 4) * 12}[path],
                         "compute": None if not pmc else {
                             "valu_tflops_equiv": pmc.get("valu_insts_per_step", 0) * 64 * 2 / (kernel_ms * 1e-3) / 1e12,
                             "peak_tflops": VALU_PEAK_TFLOPS,
                             "frac": pmc.get("valu_insts_per_step", 0) * 64 * 2 / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                             "frac_of_measured_issue_rate": pmc.get("valu_insts_per_step", 0) / (kernel_ms * 1e-3) / 1e12 / VALU_SUSTAINED_TINST,
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
        import statistics
        out["repeats"] = {"n": len(rep_values), "values": rep_values, "mean": statistics.fmean(rep_values),
                          "std": statistics.pstdev(rep_values) if len(rep_values) > 1 else 0.0,
                          "note": "`value` is the mean over these windows (total steps / total time).  The first window is the --steps steps "
                                  "that follow the warm-up, bracketed by barrier + synchronize on both sides (host clock; `first_window`).  "
                                  "The others are further windows of the same length run back to back behind it, timed with device events "
                                  "at the window boundaries (max over launch streams and ranks) and one synchronize at the end"}
        if segments:
            seg_rates = [n_local * 100 / (ms * 1e-3) for ms in segments]
            out["sustained"] = {"steps": args.steps, "env_steps_per_s": value,
                                "host_env_steps_per_s": n_local * world * args.steps / joined,
                                "device_env_steps_per_s": n_local * world * args.steps / (1e-3 * sum(segments)),
                                "device_wide_sync_env_steps_per_s": n_local * world * args.steps / (joined + drain[0]),
                                "prefetch_drain_s": drain[0],
                                "per_100_steps_env_steps_per_s": seg_rates,
                                "min": min(seg_rates), "max": max(seg_rates),
                                "note": "one window long enough for every env to pass its time limit (auto-reset inside); device time "
                                        "of each 100-step slice on rank 0, the way the reference logs its step time (run_eval.py:103-124). "
                                        "`host_env_steps_per_s` = host clock from the barrier to the moment the launch streams are "
                                        "joined (stream.synchronize(): the K steps are complete - what a caller that synchronises its own "
                                        "stream measures, max over ranks) = `env_steps_per_s` = `value`; the contract's device-wide "
                                        "synchronize is issued right behind it and returned `prefetch_drain_s` later "
                                        "(`device_wide_sync_env_steps_per_s`): it also waits for the reset prefetch of FUTURE "
                                        "episodes that the last mass reset started on the library's low-priority stream (in a longer run that "
                                        "work overlaps the next episode's steps - the dips of `per_100_steps` - instead of standing alone at the "
                                        "end; when the wait is below 1 % of the window, as in the driver's 20-step windows, `value` is clocked at "
                                        "the device-wide synchronize itself); `device_env_steps_per_s` = the same steps over the device time of the launch streams"}
        if hasattr(envs[0], "sim") and hasattr(envs[0].sim, "info"):
            out["config"]["step_path"] = envs[0].sim.info()
            if path == 2:
                out["config"]["chain_stats"] = envs[0].sim.chain_stats()
        if not args.no_cpu_baseline and on_gpu:
            # world > 1 as well: rank 0 alone, after the timed region and after every collective of the run - the other
            # ranks are not held at a barrier for it, they finalize and exit while rank 0 times the oracle on the host cores
            if world > 1:
                sdist.finalize()
            out["cpu_baseline"] = cpu_baseline()
            if world > 1:
                out["cpu_baseline"]["note_ranks"] = f"timed on rank 0 of {world} after the GPU windows, the other ranks idle or gone"
        print(json.dumps(out))
    sdist.finalize()


if __name__ == "__main__":
    main()
