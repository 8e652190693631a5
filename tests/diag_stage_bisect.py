"""Diagnostic (lives under tests/ because it uses the oracle as the checker): stage-by-stage comparison of the engine
against the oracle, run on the GPU box.
    python tests/diag_stage_bisect.py [W] [B]
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

print("---- vs f16-emulating oracle ----")
taps16 = {}
ref16 = hctr_ref.forward_f16(sd, x, taps16).numpy()
model(imgs, widths=widths)
for name in ("conv0_1", "stage0", "stage1", "stage2", "stage3", "stage4"):
    a = model.debug_activation(name, B)
    r = taps16[name].numpy()
    err = np.abs(a - r)
    nz = err > 0
    print("%-8s max|ref| %8.3f  max err %9.5f  mean err %.6f  frac differing %.4f  max rel(of differing) %.2e" %
          (name, np.abs(r).max(), err.max(), err.mean(), nz.mean(),
           (err[nz] / np.maximum(np.abs(r[nz]), 1e-3)).max() if nz.any() else 0), flush=True)
err = np.abs(got - ref16)
print("logits   max|ref| %8.3f  max err %8.4f  mean err %.5f  argmax agree %.4f" %
      (np.abs(ref16).max(), err.max(), err.mean(), (got.argmax(2) == ref16.argmax(2)).mean()))

print("---- stage-1 block buffers vs f16-emulating oracle ----")
# after the forward: p1.1 = block1.0 output, p1.2 = block1.1.conv1 output, p1.0 = block1.1 output
for buf, tap in (("p1.1", "block1.0"), ("p1.2", "block1.1.conv1"), ("p1.0", "block1.1")):
    a = model.debug_activation(buf, B)
    r = taps16[tap].numpy()
    err = np.abs(a - r)
    print("%-6s = %-16s max|ref| %7.3f max err %9.5f mean err %.6f frac differing %.4f" %
          (buf, tap, np.abs(r).max(), err.max(), err.mean(), (err > 0).mean()), flush=True)
# variant: what if the residual/downsample or the SE scale were different? test residual alone:
a = model.debug_activation("p1.1", B)
o = taps16["block1.0.conv2"].numpy(); sc = taps16["block1.0.scale"].numpy(); r = taps16["block1.0.res"].numpy()
for label, val in (("o*sc+r (mul,add)", np.maximum(o * sc[:, :, None, None] + r, 0)),
                   ("fma in f64 then f32", np.maximum((o.astype(np.float64) * sc[:, :, None, None] + r), 0).astype(np.float32))):
    v16 = val.astype(np.float16).astype(np.float32)
    print("  variant %-22s frac differing %.4f" % (label, (np.abs(a - v16) > 0).mean()))
