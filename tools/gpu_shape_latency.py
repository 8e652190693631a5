"""Cost of a NEW (lines, width) shape for the engine (workspace set-up) vs a repeated one:

    python tools/gpu_shape_latency.py [B]

Ragged workloads (test.py with arbitrary line widths, bucketing.plan_batches) present a different padded width
with almost every batch; this prints the first-call and repeat-call time per shape and the lines/s of a pass
over 24 distinct widths."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import hctr_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
synth = hctr_amd.synth
C = synth.DEFAULT_VOCAB + 2
m = hctr_amd.hctr_model(C).cuda(0)
m.load_state_dict(synth.make_state_dict(C, seed=0))
widths = [1500 + 37 * i for i in range(24)]
imgs = {w: synth.make_line_images(B, w, 3) for w in widths}
m.greedy(imgs[widths[0]][:1, :, :64])             # library / kernels warm
first, again = [], []
t_all = time.perf_counter()
for w in widths:
    t0 = time.perf_counter(); m.greedy(imgs[w]); first.append(time.perf_counter() - t0)
t_all = time.perf_counter() - t_all
for w in widths:
    t0 = time.perf_counter(); m.greedy(imgs[w]); again.append(time.perf_counter() - t0)
cols = B * np.array(widths)
print("B=%d: first call per new width: median %.1f ms (min %.1f max %.1f); repeat call: median %.1f ms"
      % (B, 1e3 * np.median(first), 1e3 * min(first), 1e3 * max(first), 1e3 * np.median(again)))
print("pass over 24 new widths: %.1f lines/s; repeat pass: %.1f lines/s" % (B * len(widths) / t_all, B * len(widths) / sum(again)))
