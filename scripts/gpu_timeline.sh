#!/bin/bash
# kernel timeline of a few control steps (rocprofv3 --kernel-trace): start/end/grid per dispatch
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_t; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline $BENCH_ARGS > $O/timeline.log 2>&1
f=$(find /tmp/prof_t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("k_")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last full step: find the last two k_order dispatches
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_order")]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    print("%-14s grid %7s  start %8.1f us  dur %7.1f us  stream/queue %s" % (r["Kernel_Name"].split("(")[0], r.get("Grid_Size", r.get("Grid_Size_X", "?")), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?")))
print("step span %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
PY
