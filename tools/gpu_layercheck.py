"""Per-layer tight check details (GPU box): engine tap -> oracle next layer vs engine next tap."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hctr_amd
from oracle import hctr_ref
synth = hctr_amd.synth
C = synth.DEFAULT_VOCAB + 2
sd = synth.make_state_dict(C, seed=0)
widths = [130, 130]; B = 2
imgs = synth.make_line_images(B, 130, 24)
model = hctr_amd.hctr_model(C).cuda(0); model.load_state_dict(sd)
model(imgs, widths=widths)
e01 = model.debug_activation("conv0_1", B)
_, s0 = hctr_ref._conv_f16(sd, torch.from_numpy(e01), "cnn.conv0_2", "cnn.bn0_2", True, 1, pool=True)
r = s0.numpy(); a = model.debug_activation("stage0", B)
diff = np.abs(a - r); ulp = np.maximum(np.abs(r) * 2.0 ** -10, 3e-5)
bad = diff > ulp
print("conv0_2: bad %d of %d; frac differing %.4f" % (bad.sum(), bad.size, (diff > 0).mean()))
idx = np.argsort(-(diff / ulp).ravel())[:12]
for i in idx:
    print("  ref %.6f got %.6f diff %.3e (%.2f ulp)" % (r.ravel()[i], a.ravel()[i], diff.ravel()[i], (diff / ulp).ravel()[i]))
sub = (e01 > 0) & (e01 < 6.2e-5)
print("conv0_1 outputs that are fp16 subnormal: %.4f of elements" % sub.mean())
# emulate a flush-to-zero of subnormal fp16 inputs and compare again
e01f = np.where(sub, 0.0, e01).astype(np.float32)
_, s0f = hctr_ref._conv_f16(sd, torch.from_numpy(e01f), "cnn.conv0_2", "cnn.bn0_2", True, 1, pool=True)
rf = s0f.numpy(); d2 = np.abs(a - rf); u2 = np.maximum(np.abs(rf) * 2.0 ** -10, 3e-5)
print("with subnormal inputs flushed in the oracle: bad %d; frac differing %.4f" % ((d2 > u2).sum(), (d2 > 0).mean()))

print("---- locations and float64 truth ----")
w, bias = hctr_ref._fold(sd, "cnn.conv0_2", "cnn.bn0_2", True)
x64 = torch.from_numpy(e01).double()
y64 = torch.nn.functional.conv2d(x64, w.double(), bias.double(), padding=1)
y64 = torch.nn.functional.max_pool2d(torch.relu(y64), (2, 1), (2, 1)).numpy()
ii = np.argwhere(bad)
print("bad (b,c,h,w) sample:", ii[:16].tolist())
print("w histogram of bad:", np.bincount(ii[:, 3], minlength=130).nonzero()[0].tolist())
print("h histogram of bad:", np.bincount(ii[:, 2], minlength=64).nonzero()[0].tolist())
e_ref = np.abs(r - y64)[bad]; e_got = np.abs(a - y64)[bad]
print("on the bad elements: |oracle32 - f64| max %.3e mean %.3e ; |engine - f64| max %.3e mean %.3e" %
      (e_ref.max(), e_ref.mean(), e_got.max(), e_got.mean()))
allr = np.abs(r - y64); allg = np.abs(a - y64)
print("all elements: |oracle32 - f64| max %.3e ; |engine - f64| max %.3e" % (allr.max(), allg.max()))
