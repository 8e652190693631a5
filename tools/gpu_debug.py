"""Stage-by-stage comparison of the engine against the oracle (run on the GPU box).
    python tools/gpu_debug.py [W] [B]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hctr_amd  # noqa: E402
from oracle import hctr_ref  # noqa: E402

synth = hctr_amd.synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 67
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3
C = synth.DEFAULT_VOCAB + 2
t0 = time.time()
sd = synth.make_state_dict(C, seed=0)
print("weights %.1fs" % (time.time() - t0), flush=True)
widths = [W] + [max(1, W - 17 * (i + 1)) for i in range(B - 1)]
imgs = synth.make_line_images(B, W, seed=22)
x = synth.normalize_pad(imgs, widths)
model = hctr_amd.hctr_model(C).cuda(0)
t0 = time.time()
model.load_state_dict(sd)
print("ingest %.1fs" % (time.time() - t0), flush=True)
taps = {}
ref = hctr_ref.forward(sd, x, taps).numpy()
got = model(imgs, widths=widths)          # uint8 path: NormalizePAD on the device
for name in ("conv0_1", "stage0", "stage1", "stage2", "stage3", "stage4"):
    a = model.debug_activation(name, B)
    r = taps[name].numpy()
    err = np.abs(a - r)
    print("%-8s shape %-20s max|ref| %8.3f  max err %8.4f  mean err %.5f  rel %.2e" %
          (name, a.shape, np.abs(r).max(), err.max(), err.mean(), err.max() / max(1e-9, np.abs(r).max())), flush=True)
err = np.abs(got - ref)
print("logits   shape %-20s max|ref| %8.3f  max err %8.4f  mean err %.5f" % (got.shape, np.abs(ref).max(), err.max(), err.mean()))
srt = np.sort(ref, axis=2)
margin = srt[:, :, -1] - srt[:, :, -2]
agree = got.argmax(2) == ref.argmax(2)
print("argmax agree %.4f; disagreeing margins: %s" % (agree.mean(), np.sort(margin[~agree])[:10]))
got_f32 = model(x)
print("u8-vs-f32 input paths max diff", np.abs(got_f32 - got).max())
