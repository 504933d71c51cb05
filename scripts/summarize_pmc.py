import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    k = r.get("Kernel_Name", "")
    if "k_physics" not in k: continue
    key = (r["Dispatch_Id"], r["Counter_Name"])
    agg[key] = agg.get(key, 0.0) + float(r["Counter_Value"])
disp = sorted({k[0] for k in agg}, key=int)
for d in disp:
    print("dispatch", d, {c: v for (dd, c), v in agg.items() if dd == d})
