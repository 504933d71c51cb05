#!/bin/bash
# kernel timeline of one control step (rocprofv3 --kernel-trace): per launch chain (queue) the sum of kernel durations and
# of the gaps between consecutive dispatches, then the dispatches of the last full step.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_t; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline $BENCH_ARGS > $O/timeline.log 2>&1
f=$(find /tmp/prof_t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("k_")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_order")]
def analyse(a, b, verbose):
    t0 = int(rows[a]["Start_Timestamp"])
    span = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
    byq = collections.defaultdict(list)
    for r in rows[a:b]:
        byq[r.get("Queue_Id", "?")].append(r)
    out = []
    for q, rs in sorted(byq.items()):
        rs.sort(key=lambda r: int(r["Start_Timestamp"]))
        dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
        gaps = [(int(rs[i + 1]["Start_Timestamp"]) - int(rs[i]["End_Timestamp"])) / 1e3 for i in range(len(rs) - 1)]
        first = (int(rs[0]["Start_Timestamp"]) - t0) / 1e3; last = (int(rs[-1]["End_Timestamp"]) - t0) / 1e3
        out.append((q, len(rs), first, last, dur, sum(gaps), max(gaps) if gaps else 0.0))
    return span, out
spans = []
for k in range(3, len(idx) - 1):
    span, out = analyse(idx[k], idx[k + 1], False)
    spans.append(span)
print("step spans (us): mean %.0f min %.0f max %.0f over %d steps" % (sum(spans) / len(spans), min(spans), max(spans), len(spans)))
span, out = analyse(idx[-3], idx[-2], True)
print("last full step: span %.1f us" % span)
for q, n, first, last, dur, gsum, gmax in out:
    print("  queue %s: %2d dispatches, first start %7.1f, last end %7.1f, kernel time %7.1f, gaps %6.1f (max %5.1f)" % (q, n, first, last, dur, gsum, gmax))
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    print("%-14s grid %7s  start %8.1f us  dur %7.1f us  queue %s" % (r["Kernel_Name"].split("(")[0], r.get("Grid_Size", r.get("Grid_Size_X", "?")), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?")))
PY
