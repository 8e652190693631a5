import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hctr_amd
synth = hctr_amd.synth
C = synth.DEFAULT_VOCAB + 2
sd = synth.make_state_dict(C, seed=0)
B = 2
imgs = synth.make_line_images(B, 130, 24)
model = hctr_amd.hctr_model(C).cuda(0); model.load_state_dict(sd)
model(imgs)
out = {n.replace(".", "_"): model.debug_activation(n, B).astype(np.float16) for n in ("conv0_1", "stage0", "p1.1", "p1.2")}
model(imgs)
out["stage0_run2"] = model.debug_activation("stage0", B).astype(np.float16)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "taps.npz"), **out)
print("saved", {k: v.shape for k, v in out.items()})
