"""Averages rocprofv3 --pmc counter_collection.csv per kernel: python summarize_pmc.py DIR [kernel-substring ...]
Only the second half of each kernel's dispatches is used (steady state of the rollout)."""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
want = sys.argv[2:] or ["k_"]
per = collections.defaultdict(lambda: collections.defaultdict(dict))      # kernel -> counter -> dispatch -> value
for r in rows:
    k = r.get("Kernel_Name", "")
    if not any(w in k for w in want):
        continue
    name = k.split("(")[0]
    d = per[name][r["Counter_Name"]]
    d[int(r["Dispatch_Id"])] = d.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
for name, ctrs in per.items():
    out = {}
    for c, d in ctrs.items():
        ids = sorted(d)
        ids = ids[len(ids) // 2:]
        out[c] = sum(d[i] for i in ids) / max(1, len(ids))
    print(name, "dispatches", len(next(iter(ctrs.values()))), {c: (round(v, 1)) for c, v in out.items()})
