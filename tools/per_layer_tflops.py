"""profiles/<tag>_per_layer.json (tools/summarize_prof.py) -> a text table with the algorithmic TFLOP/s of every launch.
    python tools/per_layer_tflops.py profiles/r03_per_layer.json [columns=128000] > profiles/r03_per_layer_tflops.txt"""
import json
import sys

MFLOP = {"stem+conv0_2+pool": 9.44 + 0.147, "block1.0.conv1": 9.44, "block1.0.conv2": 18.87 + 1.05, "block1.1.conv1": 18.87,
         "block1.1.conv2": 18.87, "conv1+pool": 18.87, "block2.0.conv1": 18.87, "block2.0.conv2": 37.75 + 2.10,
         "conv2+pool": 37.75, "block3.0.conv1": 37.75, "block3.0.conv2": 75.50 + 4.19, "conv3+pool": 75.50,
         "block4.0.conv1": 37.75, "block4.0.conv2": 37.75, "conv4+pool": 37.75, "head.linear": 30.14}
for s, n in ((2, 4), (3, 5)):
    for i in range(1, n):
        MFLOP["block%d.%d.conv1" % (s, i)] = 37.75 if s == 2 else 75.50
        MFLOP["block%d.%d.conv2" % (s, i)] = 37.75 if s == 2 else 75.50


def flops(layer):
    base = layer.split("+")[0] if layer.startswith(("block", "head")) else layer
    return MFLOP.get(base, MFLOP.get(layer))


rows = json.load(open(sys.argv[1]))
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 128000
print("# rocprofv3 kernel trace of bench.py (config 2: %d pixel columns per launch), last timed step; PMC passes separate" % cols)
print("# layer                          ms   algorithmic TFLOP/s   FETCH GB (x2-corrected)   WRITE GB   MFMA busy %   clock GHz")
tot = 0.0
groups = {}
for r in rows:
    f = flops(r["layer"])
    tf = "%9.1f" % (f * 1e6 * cols / (r["ms"] * 1e-3) / 1e12) if f else " " * 9
    tot += r["ms"]
    g = "stem" if r["layer"].startswith("stem") else ("head" if r["layer"].startswith(("head", "argmax", "ctc")) else
         "stage %s" % (r["layer"][5] if r["layer"].startswith("block") else r["layer"][4]))
    groups[g] = groups.get(g, 0.0) + r["ms"]
    fx = "%10.3f" % r["fetch_gb_x2"] if r.get("fetch_gb_x2") is not None else " " * 10
    wr = "%10.3f" % r["write_gb"] if r.get("write_gb") is not None else " " * 10
    bz = "%8.1f" % r["mfma_busy_pct"] if r.get("mfma_busy_pct") is not None else " " * 8
    ck = "%8.3f" % r["clock_ghz"] if r.get("clock_ghz") is not None else " " * 8
    print("%-28s %9.3f   %s %s %s %s %s" % (r["layer"], r["ms"], tf, fx, wr, bz, ck))
print("# sum of kernels %.3f ms;  %s" % (tot, "  ".join("%s %.2f ms" % kv for kv in groups.items())))
