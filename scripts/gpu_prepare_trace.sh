#!/bin/bash
# What the stepping kernels do while the reset prefetch (k_prepare) refills the cache after a mass reset: kernel trace
# of 560 control steps, then the dispatches of the 60 ms after the long k_prepare starts.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_p; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_p -- python3 $R/scripts/gpu_step_trace.py 4096 560 stream > $O/prepare_trace.log 2>&1
f=$(find /tmp/prof_p -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
for r in rows: r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"]); r["n"] = r["Kernel_Name"].split("(")[0][-24:]
rows.sort(key=lambda r: r["s"])
prep = [r for r in rows if "k_prepare" in r["n"] and r["e"] - r["s"] > 3e8]
print("long k_prepare runs:", [(round((r["e"] - r["s"]) / 1e6), r.get("Queue_Id")) for r in prep])
p = prep[-1]
t0 = p["s"]
print("k_prepare start 0, end %.1f ms, queue %s" % ((p["e"] - t0) / 1e6, p.get("Queue_Id")))
orders = [r for r in rows if r["n"].startswith("k_order")]
print("control-step spans (k_order to k_order) around the refill, ms:")
for a_, b_ in zip(orders, orders[1:]):
    if t0 - 3e7 <= a_["s"] <= p["e"] + 3e7:
        inside = [r for r in rows if a_["s"] <= r["s"] < b_["s"] and r["n"].startswith("k_pipe")]
        longest = max(inside, key=lambda r: r["e"] - r["s"]) if inside else None
        print("  at %8.1f ms: span %7.2f ms; longest kernel %s %.2f ms (queue %s)" % ((a_["s"] - t0) / 1e6, (b_["s"] - a_["s"]) / 1e6,
              longest["n"] if longest else "-", (longest["e"] - longest["s"]) / 1e6 if longest else 0, longest.get("Queue_Id") if longest else "-"))
PY
