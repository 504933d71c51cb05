#!/bin/bash
# Round report on the GPU box (run through gpurun): GPU tests, smoke, the bench lines of every workload, then the profile
# of scripts/gpu_profile.sh (rocprofv3 kernel stats + separate PMC passes).  Everything lands in gpurun_out/<tag>_*; the
# files to keep are copied into profiles/ by hand.
#   usage: bash scripts/gpu_round_report.sh r02
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -4 > $O/${TAG}_clocks_before.txt
# the driver's command first, on the cold box
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/${TAG}_bench_driver_args.json
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/${TAG}_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 > $O/${TAG}_smoke.txt
timeout 600 python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_100.json
timeout 900 python bench.py --steps 500 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_500.json
timeout 900 python bench.py --envs-per-gpu 16384 --steps 40 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_16384.json
timeout 900 python bench.py --envs-per-gpu 32768 --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_32768.json
timeout 900 python bench.py --workload pickplace --steps 40 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_pickplace.json
timeout 900 python bench.py --workload mixed --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_mixed.json
timeout 600 python bench.py --fused --no-cpu-baseline --steps 40 2>/dev/null | tail -1 > $O/${TAG}_bench_fused.json
# the other step paths (same device functions, bit-identical): per-env chained persistent kernel, merged launches
timeout 600 python bench.py --pipeline 2 --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/${TAG}_bench_chained.json
timeout 600 python bench.py --pipeline 3 --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/${TAG}_bench_merged.json
timeout 900 python bench.py --steps 1500 --warmup 10 --no-cpu-baseline --repeats 1 2>/dev/null | tail -1 > $O/${TAG}_bench_1500.json
timeout 600 python scripts/gpu_single_env_latency.py 2>/dev/null | tail -1 > $O/${TAG}_single_env_latency.json
timeout 1200 python scripts/gpu_soak_rates.py 2>/dev/null | grep "^{" > $O/${TAG}_soak_rates.json
timeout 600 python scripts/gpu_reset_cost.py 2>&1 | tail -6 > $O/${TAG}_reset_cost.txt
# ALOHA hand-over on the general-tree engine: throughput at three batch sizes, stage times, kernel stats
timeout 600 python bench.py --workload aloha --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/${TAG}_bench_aloha.json
timeout 900 python bench.py --workload aloha --steps 600 --warmup 5 2>/dev/null | tail -1 > $O/${TAG}_bench_aloha_600.json
timeout 600 python bench.py --workload dining --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/${TAG}_bench_dining.json
timeout 600 python bench.py --workload dining --envs-per-gpu 4096 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/${TAG}_bench_dining_4096.json
timeout 600 python bench.py --narrowphase mpr --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_mpr_option.json
timeout 600 python scripts/gpu_aloha_bench.py banana 2>/dev/null | grep workload > $O/${TAG}_aloha_bench.json
timeout 600 python scripts/gpu_aloha_bench.py pen 2>/dev/null | grep workload > $O/${TAG}_aloha_bench_pen.json
timeout 600 python scripts/gpu_tree_phases.py 2>/dev/null | grep mask > $O/${TAG}_aloha_phases.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_aloha && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_aloha -- python3 $R/scripts/gpu_aloha_bench.py banana > /dev/null 2>&1; find /tmp/prof_aloha -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_aloha_kernel_stats.csv \; )
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -4 > $O/${TAG}_clocks_after.txt
bash $R/scripts/gpu_profile.sh $TAG > $O/${TAG}_profile.log 2>&1
# round 6: instruction-cache counters of the step's kernels, the general-tree engine's PMC pass, the narrowphase phase clocks (profiling build)
bash $R/scripts/gpu_pmc_icache.sh > $O/${TAG}_icache.txt 2>&1
bash $R/scripts/gpu_pmc_tree.sh > $O/${TAG}_aloha_pmc.txt 2>&1
[ -f $R/so101_sim_amd/csrc/libso101_hip_clocks.so ] && SO101_HIP_LIB=$R/so101_sim_amd/csrc/libso101_hip_clocks.so TICKS_STEPS=60 timeout 600 python scripts/gpu_narrow_ticks.py > $O/${TAG}_narrow_ticks.txt 2>&1
[ -f $R/so101_sim_amd/csrc/libso101_hip_clocks.so ] && SO101_HIP_LIB=$R/so101_sim_amd/csrc/libso101_hip_clocks.so timeout 600 python scripts/gpu_solve_stages.py > $O/${TAG}_solve_stages.txt 2>&1
for f in $O/${TAG}_pytest_gpu.txt $O/${TAG}_smoke.txt $O/${TAG}_reset_cost.txt; do echo "== $f"; cat $f; done
for f in $O/${TAG}_bench_*.json; do echo "== $(basename $f): $(cut -c1-110 $f)"; done
tail -8 $O/${TAG}_profile.log
