#!/bin/bash
# SQ counters of k_narrow / k_pipe_solve (rocprofv3 --pmc, kernels serialised by the profiler) for the default library and ab/ variants
#   scripts/gpu_pmc_ab.sh <tag> [variant ...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in default "$@"; do
  if [ "$n" = default ]; then unset SO101_HIP_LIB; else export SO101_HIP_LIB=$R/ab/lib_$n.so; fi
  for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    k=$(echo $c | cut -d' ' -f1)
    rm -rf /tmp/pm_${n}_$k
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm_${n}_$k -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-prefetch --repeats 1 > $O/${TAG}_pmc_$n.log 2>&1
  done
  echo "== $n"
  python3 - $n <<'PY'
import csv, glob, collections, sys
n = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob('/tmp/pm_%s_*/**/*counter_collection.csv' % n, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        for kk in ('k_narrow', 'k_pipe_solve', 'k_pipe_begin'):
            if k.startswith(kk):
                tot[kk][r['Counter_Name']] += float(r['Counter_Value']); cnt[(kk, r['Counter_Name'])].add(r['Dispatch_Id'])
for kk, d in tot.items():
    per = {c: v / max(1, len(cnt[(kk, c)])) for c, v in d.items()}
    w = per.get('SQ_WAVES', 1)
    print("  %-13s waves %6.0f | per wave: VALU %7.0f SALU %6.0f LDS %5.0f SMEM %4.0f VMEM_RD %4.0f VMEM_WR %4.0f | wave-cycles %8.0f  active %4.1f%%  wait_any %4.1f%%  wait_inst %4.1f%% | lanes %4.1f%%" % (
        kk, w, per.get('SQ_INSTS_VALU', 0) / w, per.get('SQ_INSTS_SALU', 0) / w, per.get('SQ_INSTS_LDS', 0) / w, per.get('SQ_INSTS_SMEM', 0) / w,
        per.get('SQ_INSTS_VMEM_RD', 0) / w, per.get('SQ_INSTS_VMEM_WR', 0) / w, 4 * per.get('SQ_WAVE_CYCLES', 0) / w,
        100 * per.get('SQ_ACTIVE_INST_ANY', 0) / max(1, per.get('SQ_WAVE_CYCLES', 1)), 100 * per.get('SQ_WAIT_ANY', 0) / max(1, per.get('SQ_WAVE_CYCLES', 1)),
        100 * per.get('SQ_WAIT_INST_ANY', 0) / max(1, per.get('SQ_WAVE_CYCLES', 1)),
        100 * per.get('SQ_THREAD_CYCLES_VALU', 0) / max(1, 64 * per.get('SQ_ACTIVE_INST_VALU', 1))))
PY
done
