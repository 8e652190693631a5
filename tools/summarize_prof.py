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
        for k in ("se_apply_kernel", "stem_kernel", "row_topk_kernel", "se_border_kernel", "se_premean_kernel",
                  "log_softmax_rows_kernel", "row_candidates_kernel"):
            if k in name:
                return k
    return name.split("(")[0][:64]


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
    # MFMA pipe utilisation (own PMC pass): SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs,
    # GRBM_GUI_ACTIVE sums the active cycles of the 8 XCDs -> busy / (128 * gui_active); clock = gui / 8 / time
    mf = collections.defaultdict(lambda: collections.defaultdict(list))
    files = glob.glob(os.path.join(d, "pmc_mfma", "*", "*_counter_collection.csv"))
    if files:
        for r in csv.DictReader(open(files[0])):
            k = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
            mf[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            mf[k]["_t_" + r["Counter_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in dur.values())
    print("| kernel | grid (threads) | launches | avg ms | % time | FETCH_SIZE/launch (GB raw / x2) | WRITE_SIZE/launch (GB) | MFMA pipe busy | clock GHz |")
    print("|---|---|---|---|---|---|---|---|---|")
    for key, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        f = pmc["fetch"].get(key)
        w = pmc["write"].get(key)
        fs = "%.3f / %.3f" % (sum(f) / len(f) * 1024 / 1e9, 2 * sum(f) / len(f) * 1024 / 1e9) if f else "-"
        wsz = "%.3f" % (sum(w) / len(w) * 1024 / 1e9) if w else "-"
        m = mf.get(key)
        util, clk = "-", "-"
        if m and m.get("GRBM_GUI_ACTIVE") and sum(m["GRBM_GUI_ACTIVE"]) > 0:
            gui = sum(m["GRBM_GUI_ACTIVE"])
            if m.get("SQ_VALU_MFMA_BUSY_CYCLES") and sum(m["SQ_VALU_MFMA_BUSY_CYCLES"]) > 0:
                util = "%.0f %%" % (100.0 * sum(m["SQ_VALU_MFMA_BUSY_CYCLES"]) / (128.0 * gui))
            clk = "%.2f" % (gui / 8.0 / sum(m["_t_GRBM_GUI_ACTIVE"]))
        print("| %s | %d | %d | %.3f | %.1f | %s | %s | %s | %s |" % (key[0], key[1], len(v), sum(v) / len(v) / 1e6,
                                                                100.0 * sum(v) / total, fs, wsz, util, clk))


def per_layer(d):
    """Label the launches of the LAST full step with the engine's layer names (launch order written by
    bench.py --layer-file) and attach each launch's PMC bytes (same position in the PMC runs)."""
    lf = os.path.join(d, "layers.txt")
    if not os.path.isfile(lf):
        return
    # (the shared-buffer workspace layout adds border-zeroing launches, several per profile entry: left out on both sides)
    names = [l.strip() for l in open(lf) if l.strip() and not l.startswith("zero_borders")]

    def engine_rows(path, key_start, key_end):
        rows = [r for r in csv.DictReader(open(path)) if "hctr" in r["Kernel_Name"] and "zero_borders" not in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r[key_start]))
        return rows

    trace = engine_rows(glob.glob(os.path.join(d, "stats", "*", "*_kernel_trace.csv"))[0], "Start_Timestamp", None)
    n = len(names)
    last = trace[-n:]
    if len(last) != n or "stem" not in last[0]["Kernel_Name"]:
        print("\n(per-layer table skipped: launch order does not line up)")
        return
    pm = {}
    for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        files = glob.glob(os.path.join(d, "pmc_" + kind, "*", "*_counter_collection.csv"))
        if files:
            rows = [r for r in csv.DictReader(open(files[0])) if "hctr" in r["Kernel_Name"] and r["Counter_Name"] == ctr
                    and "zero_borders" not in r["Kernel_Name"]]
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            pm[kind] = rows[-n:] if len(rows) >= n else None
    # matrix-pipe pass: busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 * GRBM_GUI_ACTIVE), clock = GRBM_GUI_ACTIVE / 8 / time
    mfma = {}
    files = glob.glob(os.path.join(d, "pmc_mfma", "*", "*_counter_collection.csv"))
    if files:
        allrows = [r for r in csv.DictReader(open(files[0])) if "hctr" in r["Kernel_Name"] and "zero_borders" not in r["Kernel_Name"]]
        for ctr in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            rows = [r for r in allrows if r["Counter_Name"] == ctr]
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            mfma[ctr] = rows[-n:] if len(rows) >= n else None
    print("\n## Last step, per launch\n")
    print("| layer | kernel | ms | FETCH_SIZE GB (raw / x2) | WRITE_SIZE GB | MFMA pipe busy % | clock GHz |")
    print("|---|---|---|---|---|---|---|")
    out = []
    for i, (nm, r) in enumerate(zip(names, last)):
        ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        f = float(pm["fetch"][i]["Counter_Value"]) * 1024 / 1e9 if pm.get("fetch") else None
        w = float(pm["write"][i]["Counter_Value"]) * 1024 / 1e9 if pm.get("write") else None
        busy = clk = None
        if mfma.get("GRBM_GUI_ACTIVE") and mfma.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            g = mfma["GRBM_GUI_ACTIVE"][i]
            gui = float(g["Counter_Value"])
            tns = int(g["End_Timestamp"]) - int(g["Start_Timestamp"])
            if gui > 0 and tns > 0:
                busy = 100.0 * float(mfma["SQ_VALU_MFMA_BUSY_CYCLES"][i]["Counter_Value"]) / (128.0 * gui)
                clk = gui / 8.0 / tns
        print("| %s | %s | %.3f | %s | %s | %s | %s |" % (nm, short(r["Kernel_Name"]), ms,
              "%.3f / %.3f" % (f, 2 * f) if f is not None else "-", "%.3f" % w if w is not None else "-",
              "%.1f" % busy if busy is not None else "-", "%.3f" % clk if clk is not None else "-"))
        out.append({"layer": nm, "ms": ms, "fetch_gb_x2": 2 * f if f is not None else None, "write_gb": w,
                    "mfma_busy_pct": busy, "clock_ghz": clk})
    import json
    with open(os.path.join(d, "per_layer.json"), "w") as fjs:
        json.dump(out, fjs, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
    per_layer(sys.argv[1])
