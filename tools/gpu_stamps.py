"""Phase timeline of the 3x3 conv kernel's workgroups (diagnostic kernel instance, hctr_debug_stamps).

    python tools/gpu_stamps.py [layer] [B] [W]

Runs config-2-shaped forwards with the named layer's workgroups time-stamped (s_memrealtime, 10 ns ticks) and
prints, per phase, the median / p10 / p90 duration over workgroups, and the gap between consecutive workgroups
that ran in the same CU slot (hardware ids from HW_ID / XCC_ID)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hctr_amd  # noqa: E402
_lib = hctr_amd.package._lib

layer = sys.argv[1] if len(sys.argv) > 1 else "block3.2.conv2+se"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
synth = hctr_amd.synth
C = synth.DEFAULT_VOCAB + 2
model = hctr_amd.hctr_model(C).cuda(0)
model.load_state_dict(synth.make_state_dict(C, seed=0))
imgs = synth.make_line_images(B, W, seed=2)
lib = _lib.load()
cap = 1 << 17
model.greedy(imgs)                                   # warm-up (workspace, clocks)
_lib.check(int(lib.hctr_debug_stamps(model._ctx, layer.encode(), None, cap)), model._ctx)
model.greedy(imgs)
model.greedy(imgs)
out = np.zeros((cap, 16), np.uint64)
n = int(lib.hctr_debug_stamps(model._ctx, None, out.ctypes.data_as(ctypes.c_void_p), cap))
if n <= 0:
    raise SystemExit("no stamps recorded for layer %r (n=%d)" % (layer, n))
s = out[:n].astype(np.int64)
t = s[:, :6] * 10e-3                                  # microseconds
names = ["entry->prologue issued", "prologue issued->operands landed", "operands landed->K loop done",
         "K loop done->epilogue done", "epilogue done->stores drained"]
print("layer %s: %d workgroups, span %.1f us" % (layer, n, t[:, 5].max() - t[:, 0].min()))
for i, nm in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print("  %-36s median %7.2f us   p10 %7.2f   p90 %7.2f" % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
e = s[:, [3, 8, 9, 10, 11, 4]] * 10e-3
for i, nm in enumerate(["  epilogue: bias loaded + added", "  epilogue: residual / SE scale", "  epilogue: relu, pool, rounding",
                        "  epilogue: SE partial sums", "  epilogue: output stores issued"]):
    d = e[:, i + 1] - e[:, i]
    print("  %-36s median %7.2f us   p10 %7.2f   p90 %7.2f" % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
life = t[:, 5] - t[:, 0]
print("  %-36s median %7.2f us   p10 %7.2f   p90 %7.2f" % ("workgroup lifetime", np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
hw, xcc = s[:, 6], s[:, 7] & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("  distinct CUs seen: %d" % len(np.unique(key)))
gaps, conc = [], []
for k in np.unique(key):
    idx = np.where(key == k)[0]
    order = idx[np.argsort(t[idx, 0])]
    ends = np.sort(t[idx, 5])
    # two slots per CU: a workgroup starts when one of the two residents has ended
    st = t[order, 0]
    for j in range(2, len(order)):
        prev_end = ends[j - 2]                       # the (j-1)th earliest end frees the slot for the jth start
        gaps.append(st[j] - prev_end)
    conc.append(len(idx))
gaps = np.array(gaps)
print("  slot gap (end of a workgroup -> entry of the next in that CU): median %.2f us  p10 %.2f  p90 %.2f" %
      (np.median(gaps), np.percentile(gaps, 10), np.percentile(gaps, 90)))
print("  workgroups per CU: min %d  median %d  max %d" % (min(conc), int(np.median(conc)), max(conc)))

# ---- how the two workgroups of a CU are phased: share of the layer's span during which 0 / 1 / 2 of a CU's resident
# workgroups are inside their K loop (stamps 2..3). Lockstep tiles leave the matrix pipe idle whenever both are outside.
span0, span1 = t[:, 0].min(), t[:, 5].max()
occ = np.zeros(3)
for k in np.unique(key):
    idx = np.where(key == k)[0]
    ev = [(t[i, 2], 1) for i in idx] + [(t[i, 3], -1) for i in idx]
    ev.sort()
    cur, last = 0, span0
    for tm, d in ev:
        occ[min(cur, 2)] += tm - last
        cur += d
        last = tm
    occ[min(cur, 2)] += span1 - last
occ /= occ.sum()
print("  CU time with 0 / 1 / 2 workgroups in the K loop: %.3f / %.3f / %.3f" % tuple(occ))
