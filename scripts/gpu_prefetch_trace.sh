#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp; export TMPDIR=/tmp
for n in default r04; do
  if [ "$n" = default ]; then unset SO101_HIP_LIB; else export SO101_HIP_LIB=$R/ab/lib_$n.so; fi
  rm -rf /tmp/pt_$n
  rocprofv3 --kernel-trace --output-format csv -d /tmp/pt_$n -- python3 $R/bench.py --steps 500 --warmup 10 --repeats 1 --no-cpu-baseline > /dev/null 2>&1
  f=$(find /tmp/pt_$n -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] || { echo "$n: no kernel_trace.csv (the profiler run failed)"; continue; }
  python3 - $f $n <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ko = [r for r in rows if r["Kernel_Name"].startswith("k_order")]
t0 = int(ko[10]["Start_Timestamp"]); tend = max(int(r["End_Timestamp"]) for r in rows if r["Kernel_Name"].startswith("k_pipe_solve"))
print(sys.argv[2], "timed region %.2f s (first timed k_order to last k_pipe_solve end)" % ((tend - t0) / 1e9))
for r in rows:
    if "k_prepare" in r["Kernel_Name"]:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e9, (int(r["End_Timestamp"]) - t0) / 1e9
        if e - s > 0.01: print("   k_prepare start %.3f s  end %.3f s  dur %.3f s  grid %s" % (s, e, e - s, r.get("Grid_Size", r.get("Grid_Size_X"))))
PY
done
