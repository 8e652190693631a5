"""Summarise a tools/profile_rocprof.sh output directory into one markdown table.

    python tools/summarize_prof.py gpurun_out/prof_r01 > profiles/r01_summary.md

Per (kernel, grid size): launches, average duration from the kernel trace, and HBM traffic per
launch from the two PMC passes. FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950
FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads (MI355X_MICROARCH.md, HBM), so
the read side is shown both raw and doubled.
"""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("void hctr::", "").replace("hctr::", "").replace("(hctr::ConvArgs)", "")
    if name.startswith("_ZN4hctr"):
        for k in ("se_apply_kernel", "stem_kernel", "row_topk_kernel"):
            if k in name:
                return k
    return name.split("(")[0][:44]


def main(d):
    trace = glob.glob(os.path.join(d, "stats", "*", "*_kernel_trace.csv"))[0]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        g = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
        dur[(short(r["Kernel_Name"]), g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    pmc = {}
    for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        acc = collections.defaultdict(list)
        files = glob.glob(os.path.join(d, "pmc_" + kind, "*", "*_counter_collection.csv"))
        if files:
            for r in csv.DictReader(open(files[0])):
                if r["Counter_Name"] == ctr:
                    acc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
        pmc[kind] = acc
    total = sum(sum(v) for v in dur.values())
    print("| kernel | grid (threads) | launches | avg ms | % time | FETCH_SIZE/launch (GB raw / x2) | WRITE_SIZE/launch (GB) |")
    print("|---|---|---|---|---|---|---|")
    for key, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        f = pmc["fetch"].get(key)
        w = pmc["write"].get(key)
        fs = "%.3f / %.3f" % (sum(f) / len(f) * 1024 / 1e9, 2 * sum(f) / len(f) * 1024 / 1e9) if f else "-"
        wsz = "%.3f" % (sum(w) / len(w) * 1024 / 1e9) if w else "-"
        print("| %s | %d | %d | %.3f | %.1f | %s | %s |" % (key[0], key[1], len(v), sum(v) / len(v) / 1e6,
                                                        100.0 * sum(v) / total, fs, wsz))


if __name__ == "__main__":
    main(sys.argv[1])
