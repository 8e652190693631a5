"""Diagnostic (lives under tests/ because it uses the oracle as the checker): f16x3 logits / text vs the oracle on a small
unequal-width batch.  python tests/diag_f16x3_check.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hctr_amd
from oracle import hctr_ref, ctc_ref
synth = hctr_amd.synth
C = synth.DEFAULT_VOCAB + 2
sd = synth.make_state_dict(C, seed=0)
widths = [67, 50, 33]; B = 3
imgs = synth.make_line_images(B, 67, 22)
x = synth.normalize_pad(imgs, widths)
taps = {}
ref = hctr_ref.forward(sd, x, taps).numpy()
for prec in ("f16", "f16x3"):
    m = hctr_amd.hctr_model(C, precision=prec).cuda(0); m.load_state_dict(sd)
    got = m(imgs, widths=widths)
    print("==", prec)
    for name in ("conv0_1", "stage0", "stage1", "stage2", "stage3", "stage4"):
        a = m.debug_activation(name, B); r = taps[name].numpy()
        print("  %-8s max err %.3e (scale %.2f)" % (name, np.abs(a - r).max(), np.abs(r).max()))
    e = np.abs(got - ref)
    print("  logits max err %.3e mean %.3e scale %.1f argmax agree %.4f" % (e.max(), e.mean(), np.abs(ref).max(), (got.argmax(2) == ref.argmax(2)).mean()))
    del m
imgs = synth.make_line_images(4, 2000, 2)
ref = hctr_ref.forward(sd, synth.normalize_pad(imgs)).numpy()
oc = ctc_ref.CtcCodecRef(synth.characters()); want = oc.decode(ref)
for prec in ("f16", "f16x3"):
    m = hctr_amd.hctr_model(C, precision=prec).cuda(0); m.load_state_dict(sd)
    cd = hctr_amd.ctc_codec(synth.characters()).attach(m)
    t0 = time.time(); txt = cd.labels_to_text(m.greedy(imgs)); dt = time.time() - t0
    t0 = time.time(); txt = cd.labels_to_text(m.greedy(imgs)); dt = time.time() - t0
    ed = [ctc_ref.edit_distance(a, b) for a, b in zip(txt, want)]
    print(prec, "W=2000 x4: edits", ed, "of", [len(s) for s in want], "exact lines", sum(a == b for a, b in zip(txt, want)), "time %.3fs" % dt)
    del m, cd
