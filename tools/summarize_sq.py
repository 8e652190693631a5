"""Per-layer SQ counter table from tools/profile_sq.sh output (last step of the run)."""
import collections, csv, glob, os, sys
d = sys.argv[1]
names = [l.strip() for l in open(os.path.join(d, "layers.txt")) if l.strip() and not l.startswith("zero_borders")]
n = len(names)
want = sys.argv[2].split(",") if len(sys.argv) > 2 else None
for sub in ("a", "b"):
    files = glob.glob(os.path.join(d, sub, "*", "*_counter_collection.csv"))
    if not files:
        continue
    rows = [r for r in csv.DictReader(open(files[0])) if "hctr" in r["Kernel_Name"] and "zero_borders" not in r["Kernel_Name"]]
    per = collections.OrderedDict()
    for r in rows:
        per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        per[int(r["Dispatch_Id"])]["_t"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    ids = sorted(per)[-n:]
    ctrs = sorted({k for i in ids for k in per[i] if k != "_t"})
    print("| layer | us | " + " | ".join(ctrs) + " |")
    print("|---|---|" + "---|" * len(ctrs))
    for nm, i in zip(names, ids):
        if want and not any(w in nm for w in want):
            continue
        print("| %s | %.0f | " % (nm, per[i]["_t"] / 1e3) + " | ".join("%.4g" % per[i].get(c, float("nan")) for c in ctrs) + " |")
    print()
