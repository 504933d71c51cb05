"""Regenerates the measured tables of profiles/README.md (the regions between `<!-- rNN:name -->` and `<!-- /rNN:name -->`) from the committed rNN_* files.
   python scripts/profiles_tables.py r05 <build hash>"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG, HASH = sys.argv[1], sys.argv[2]
P = os.path.join(ROOT, "profiles")
last = lambda f: json.loads(open(f).read().strip().splitlines()[-1])
g = lambda n: last(os.path.join(P, f"{TAG}_{n}.json"))
R4 = {"bench_driver_args": "725-731 k", "bench_100": "676 k", "bench_500": "588 k; device 676 k", "bench_1500": "630 k host = device", "bench_16384": "837 k",
      "bench_32768": "888 k, first 1052 k", "bench_mixed": "874 k", "bench_pickplace": "676 k", "bench_mpr_option": "819 k", "bench_fused": "254 k", "bench_chained": "249 k",
      "bench_merged": "378 k", "bench_aloha": "229 k", "bench_aloha_600": "194 k", "bench_dining": "46.6 k", "bench_dining_4096": "51.1 k"}
CMD = {"bench_driver_args": "`bench.py --gpus 1 --steps 20 --warmup 5` (what the driver runs)", "bench_100": "`--steps 100`",
       "bench_500": "`--steps 500` (every env passes its time limit once; the first window's HOST clock contains the wait for the background refill of the prefetch cache, below: {dev} k on the device clock)",
       "bench_1500": "`--steps 1500 --repeats 1` (three episodes): host clock; **{dev} k on the device clock**", "bench_16384": "`--envs-per-gpu 16384 --steps 40`",
       "bench_32768": "`--envs-per-gpu 32768 --steps 30` (per-GPU share of configs[4]; the row-pass instance of `k_narrow`)", "bench_mixed": "`--workload mixed` (configs[3])",
       "bench_pickplace": "`--workload pickplace` (configs[2], 16384 envs from the pre-grasp pool)", "bench_mpr_option": "`--narrowphase mpr` (the `-DSO101_MPR` library: no EPA, one contact per hull pair)",
       "bench_fused": "`--fused` (one k_step launch per control step)", "bench_chained": "`--pipeline 2` (per-env chained persistent kernel; `libso101_hip_exp.so`, built on demand)",
       "bench_merged": "`--pipeline 3` (merged launches; `libso101_hip_exp.so`)", "bench_aloha": "`--workload aloha` (HandOverBanana, ALOHA bimanual, general-tree engine, 4096 envs)",
       "bench_aloha_600": "... `--steps 600` with their resets", "bench_dining": "`--workload dining` (DiningPlaceBananaInBowl, 64-dof build, 1024 envs; 160 contacts / 960 rows per env now)",
       "bench_dining_4096": "... `--envs-per-gpu 4096`"}

def table():
    rows = ["| file | command | env-steps/s (`value` = mean of the windows; round 4 in brackets) | the windows (first = host-clocked) | ms / step |", "|---|---|---|---|---|"]
    for n in CMD:
        d = g(n); w = [round(x / 1e3) for x in d.get("repeats", {}).get("values", [])]
        dev = round(d.get("sustained", {}).get("device_env_steps_per_s", 0) / 1e3)
        rows.append(f"| `{TAG}_{n}.json` | {CMD[n].format(dev=dev)} | **{d['value'] / 1e3:.1f} k** ({R4[n]}) | {' / '.join(map(str, w)) + ' k' if w else ''} | {d['ms_per_step']:.2f} |")
    sg = g("single_env_latency")["launch_chains"]
    rows.append(f"| `{TAG}_single_env_latency.json` | `scripts/gpu_single_env_latency.py`: BASELINE configs[0], ONE env through `SingleEnvironment.step()`, a 500-step random-action episode | "
                f"{sg['env_steps_per_s']:.0f} env-steps/s: **{sg['step_ms_mean']:.2f} ms per `step()`** (p50 {sg['step_ms_p50']:.2f}, p99 {sg['step_ms_p99']:.2f}; round 4: 2.23 / 2.25 / 5.8), "
                f"episode {sg['episode_wall_s']:.2f} s, reset {sg['reset_s']:.2f} s | | |")
    cb = g("bench_driver_args")["cpu_baseline"]
    passed = re.search(r"(\d+) passed", open(os.path.join(P, f"{TAG}_pytest_gpu.txt")).read()).group(1)
    reset = re.search(r"placement \+ settle[^:]*: ([\d.]+) ms", open(os.path.join(P, f"{TAG}_reset_cost.txt")).read()).group(1)
    rows += ["", f"**cpu_baseline** of the driver line (fp64 oracle with its default narrowphase - EPA, hull patches -, {cb['cores']} usable cores of the box, 64 envs per thread): "
             f"{cb['value'] / 1e3:.1f} k env-steps/s on {cb['cores']} threads, {cb['single_core_value'] / 1e3:.2f} k on one.  `{TAG}_pytest_gpu.txt`: the full `-m gpu` suite of that call - "
             f"**{passed} passed** (round 4: 83).  `{TAG}_smoke.txt`: smoke OK.  `{TAG}_reset_cost.txt`: reset of 4096 envs {float(reset):.0f} ms with placement + settle, 0.27 ms from the settled-state store.  "
             f"`{TAG}_kernel_resources.txt`: VGPRs / spills / scratch / LDS of every kernel, read from the code objects (`scripts/kernel_resources.py`).  `{TAG}_experiments.txt`: the A/B lines of the round's experiments."]
    return "\n".join(rows)

def kernels():
    st = {r["Name"].split("(")[0]: r for r in csv.DictReader(open(os.path.join(P, f"{TAG}_kernel_stats.csv")))}
    r4 = {"k_narrow": "265.4 µs", "k_pipe_solve": "256.6 µs", "k_pipe_begin": "90.3 µs", "k_order": "6.4 µs"}
    out = ["| kernel | calls | average | min - max | round 4 |", "|---|---|---|---|---|"]
    for name, key in (("void k_narrow<false>", "k_narrow"), ("k_pipe_solve", "k_pipe_solve"), ("k_pipe_begin", "k_pipe_begin"), ("k_order", "k_order")):
        r = st[name]
        out.append(f"| `{name.replace('void ', '')}` | {r['Calls']} | **{float(r['AverageNs']) / 1e3:.1f} µs** | {float(r['MinNs']) / 1e3:.0f} - {float(r['MaxNs']) / 1e3:.0f} | {r4[key]} |")
    return "\n".join(out)

def counters():
    pmc = json.load(open(os.path.join(P, f"pmc_{HASH}.json"))); pk = pmc["per_kernel_per_dispatch"]
    t = lambda k: (2 * pk[k]["FETCH_SIZE"] + pk[k]["WRITE_SIZE"]) * 40 / 1e3
    return "\n".join([
        "| | `k_narrow` (40 launches) | `k_pipe_solve` (40) | step total | round 4 |", "|---|---|---|---|---|",
        f"| VALU wave-instructions | {pk['k_narrow']['SQ_INSTS_VALU'] * 40 / 1e6:.0f} M | {pk['k_pipe_solve']['SQ_INSTS_VALU'] * 40 / 1e6:.0f} M | **{pmc['valu_insts_per_step'] / 1e9:.2f} G** | 1.56 G (910 / 635 M) |",
        f"| SALU | | | {pmc['salu_insts_per_step'] / 1e6:.0f} M | 494 M |",
        f"| wave-cycles waiting | {100 * pk['k_narrow']['SQ_WAIT_ANY'] / pk['k_narrow']['SQ_WAVE_CYCLES']:.0f} % | {100 * pk['k_pipe_solve']['SQ_WAIT_ANY'] / pk['k_pipe_solve']['SQ_WAVE_CYCLES']:.0f} % | {100 * pmc['wait_fraction']:.0f} % | 48 % |",
        f"| active VALU lanes | | | {100 * pmc['active_lane_fraction']:.0f} % | 73 % |",
        f"| HBM traffic (2 x FETCH_SIZE + WRITE_SIZE) | {t('k_narrow'):.0f} MB | {t('k_pipe_solve'):.0f} MB | **{pmc['hbm_bytes_per_step'] / 1e6:.0f} MB** | 1479 MB (1010 / 411 MB) |"])

def sustained():
    tbl = ["| library | steps | host clock | device clock | per 100 steps (device) | events per env-step |", "|---|---|---|---|---|---|"]
    for l in open(os.path.join(P, f"{TAG}_sustained.txt")).read().strip().splitlines():
        m = re.match(r"(\S+)\s+steps\s+(\d+)\s+host\s+([\d.]+) k\s+device\s+([\d.]+) k\s+per100 (\[.*?\])\s+events (.*)", l)
        tbl.append(f"| {'round 5' if m.group(1) == 'default' else 'round 4 (`5a7cd8d`)'} | {m.group(2)} | {m.group(3)} k | {m.group(4)} k | {m.group(5)} | `{m.group(6)}` |")
    return "\n".join(tbl)

path = os.path.join(P, "README.md")
s = open(path).read()
for name, fn in (("table", table), ("kernels", kernels), ("counters", counters), ("sustained", sustained)):
    a, b = f"<!-- {TAG}:{name} -->", f"<!-- /{TAG}:{name} -->"
    i, j = s.index(a), s.index(b)
    s = s[:i + len(a)] + "\n" + fn() + "\n" + s[j:]
s = re.sub(r"(## Round 5 — final build `)[0-9a-f]{16}(` \(`r05_\*`, `pmc_)[0-9a-f]{16}(\.json`, `pmc_tree_)[0-9a-f]{16}", lambda m: m.group(1) + HASH + m.group(2) + HASH + m.group(3) + HASH, s)
open(path, "w").write(s)
print("profiles/README.md regenerated for", TAG, HASH)
