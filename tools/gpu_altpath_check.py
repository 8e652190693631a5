"""One forward of the engine under whatever HCTR_* kernel-selection variables are set in the environment,
checked against the committed reference fixture (tests/golden/model_small.npz). Kernel selection is read once
per process, so tests/test_gpu_parity.py::test_alternative_kernel_paths runs this script once per variant.
Exit code 0 = within the same tolerances as test_forward_matches_reference_fixture."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hctr_amd  # noqa: E402

synth = hctr_amd.synth
g = np.load(os.path.join(ROOT, "tests", "golden", "model_small.npz"))
C = synth.DEFAULT_VOCAB + 2
model = hctr_amd.hctr_model(C).cuda(0)
model.load_state_dict(synth.make_state_dict(C, seed=0))
worst = 0.0
for name, seed, widths in (("b3w67u", 22, [67, 50, 33]), ("b2w96", 23, [96, 96])):
    imgs = synth.make_line_images(len(widths), max(widths), seed)
    got = model(imgs, widths=widths)
    ref_sub = g[name + "/logits_sub"]
    tol = 0.01 * float(np.abs(ref_sub).max()) + 0.05
    err = float(np.abs(got[:, :, g["sub_classes"]] - ref_sub).max())
    agree = float((got.argmax(axis=2) == g[name + "/argmax"].astype(np.int64)).mean())
    print("%s err %.4f tol %.4f agree %.4f" % (name, err, tol, agree), flush=True)
    if not (err <= tol and agree >= 0.95):
        sys.exit(1)
    worst = max(worst, err)
    # greedy through the fused path must equal the decode of these logits
    labels = model.greedy(imgs, widths=widths)
    arg = got.argmax(axis=2).T
    for b in range(len(widths)):
        t = arg[b]
        keep = (t != 0) & (t != C - 1) & np.concatenate(([True], t[1:] != t[:-1]))
        if list(t[keep]) != list(labels[b]):
            print("greedy mismatch on line", b)
            sys.exit(2)
# fused beam front end (head GEMM twice with reducing epilogues) against the stored-logits kernels on this variant's own
# logits: identical top-k, log-probs and candidate lists (covers the part count of the 128x128 head tile, HCTR_BIG_TILES=0)
import importlib  # noqa: E402
mm = importlib.import_module(hctr_amd.package.__name__ + ".model")
imgs = synth.make_line_images(3, 120, 5)
fe = model.beam_frontend(imgs, k=10, want_candidates=True)
lg = np.ascontiguousarray(model(imgs))
ref = mm.beam_frontend_call(model._ctx, None, 0, 0, None, lg, 0, 3, 120, C, 10, True)
n = int(fe["cand_off"][-1])
same = all(np.array_equal(fe[k], ref[k]) for k in ("topk_idx", "topk_logp", "blank_logp", "cand_off")) and n > 0 and \
    np.array_equal(fe["cand_idx"][:n], ref["cand_idx"][:n]) and np.array_equal(fe["cand_logp"][:n], ref["cand_logp"][:n])
print("beam front end fused == stored logits:", same, flush=True)
if not same:
    sys.exit(3)
print("ok", worst)
