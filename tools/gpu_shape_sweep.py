"""Cross-check of two independent kernel families over many shapes (run on the GPU box):
    python tools/gpu_shape_sweep.py dump <file.npz>     (under whatever HCTR_* variables are set)
    python tools/gpu_shape_sweep.py compare a.npz b.npz
    python tools/gpu_shape_sweep.py identical a.npz b.npz      (two builds that must agree bit for bit)
`dump` stores, for ~30 random (lines, width, per-line widths) cases, the logits' per-column max / argmax and a
class subsample; `compare` requires the two runs (e.g. default halo kernels vs HCTR_HALO=0 generic kernels with
HCTR_FUSE_SE=0 HCTR_FUSE_DS=0 HCTR_FUSE_ARGMAX=0) to agree within fp16-pipeline noise on every case."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cases():
    rng = np.random.default_rng(2024)
    out = []
    for i in range(30):
        B = int(rng.integers(1, 7))
        W = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(40, 700)), 16 * int(rng.integers(1, 30)),
                            32 * int(rng.integers(1, 15)) + int(rng.integers(-1, 2))]))
        W = max(1, W)
        widths = [W] + [int(rng.integers(1, W + 1)) for _ in range(B - 1)]
        out.append((i, B, W, widths))
    return out


def dump(path):
    import hctr_amd
    s = hctr_amd.synth
    C = s.DEFAULT_VOCAB + 2
    m = hctr_amd.hctr_model(C).cuda(0)
    m.load_state_dict(s.make_state_dict(C, seed=0))
    d = {}
    for i, B, W, widths in cases():
        imgs = s.make_line_images(B, W, seed=3000 + i)
        lg = m(imgs, widths=widths)
        lab = m.greedy(imgs, widths=widths)
        d["%d/max" % i] = lg.max(axis=2)
        d["%d/arg" % i] = lg.argmax(axis=2).astype(np.int32)
        d["%d/sub" % i] = lg[:, :, ::97].copy()
        d["%d/len" % i] = np.array([len(x) for x in lab], np.int32)
        d["%d/lab" % i] = np.concatenate(lab) if sum(len(x) for x in lab) else np.zeros((0,), np.int32)
        # fused greedy must equal the decode of this run's own logits
        arg = lg.argmax(axis=2).T
        for b in range(B):
            t = arg[b]
            keep = (t != 0) & (t != C - 1) & np.concatenate(([True], t[1:] != t[:-1]))
            assert list(t[keep]) == list(lab[b]), ("fused greedy != decode of own logits", i, b)
    np.savez_compressed(path, **d)
    print("dumped", len(cases()), "cases ->", path)


def compare(pa, pb):
    a, b = np.load(pa), np.load(pb)
    worst, agree_min = 0.0, 1.0
    for i, B, W, widths in cases():
        scale = float(np.abs(a["%d/max" % i]).max())
        err = float(np.abs(a["%d/sub" % i] - b["%d/sub" % i]).max())
        agree = float((a["%d/arg" % i] == b["%d/arg" % i]).mean())
        worst, agree_min = max(worst, err / scale), min(agree_min, agree)
        assert err <= 0.012 * scale + 0.05, (i, B, W, err, scale)
        assert agree >= 0.93, (i, B, W, agree)
    print("ok: %d cases, worst relative logit difference %.4f, minimum argmax agreement %.3f" %
          (len(cases()), worst, agree_min))


def identical(pa, pb):
    """bit-for-bit equality of two dumps (two builds of the SAME arithmetic, e.g. a re-scheduled epilogue)"""
    a, b = np.load(pa), np.load(pb)
    assert sorted(a.files) == sorted(b.files)
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    assert not bad, bad[:8]
    print("identical: %d arrays of %d cases" % (len(a.files), len(cases())))


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2])
    elif sys.argv[1] == "identical":
        identical(sys.argv[2], sys.argv[3])
    else:
        compare(sys.argv[2], sys.argv[3])
