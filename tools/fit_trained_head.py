"""Fit the "trained-like" classifier head of the synthetic checkpoint (synth.make_state_dict(head="trained")).

The trunk stays the seed-0 random trunk; only ``linear`` is fitted, as a weighted ridge-regression read-out of the
trunk's 2048 features per pixel column on glyph-font lines (synth.make_font_lines): columns in the core of a glyph
regress to that glyph's class, all other columns to <blank> (glyphs cut by the right edge are left out). The result is
a checkpoint whose logits on font lines are peaky like a trained CTC model's (top-2 margins far above any
reduced-precision error), which the random head cannot offer. Output: synth_head_trained.npz (fp16 rows, ~2 MB),
committed next to synth_bn_calib.npz; the fixtures for it come from the REAL reference (make_golden_c2.py
--checkpoint trained).

    python tools/fit_trained_head.py --train-lines 768 --classes 8      (GPU box; trunk features from the engine in f16x3 mode)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hctr_amd  # noqa: E402

synth = hctr_amd.synth
TRAIN_SEED = 1001
CORE = 0.30                         # glyph box fractions: [CORE, 1-CORE) regresses to the class, the rest to <blank>
                                    # (no don't-care band: unconstrained transition columns decode as stray characters)
ALPHA = 30.0                        # logit scale of the fitted read-out (regression targets are 0/1)


def column_targets(boxes, width):
    """per column: class index 1..K (glyph core), 0 (blank) or -1 (ignored: cut-off glyph)"""
    t = np.zeros((width,), np.int32)
    for k, x0, x1 in boxes:
        gw = x1 - x0
        if x1 > width:                                   # cut by the right edge: ambiguous
            t[x0:width] = -1
            continue
        t[x0 + int(CORE * gw):x1 - int(CORE * gw)] = 1 + k
    return t


class Features(object):
    def __init__(self, backend, precision="f16x3"):
        self.backend = backend
        C = synth.DEFAULT_VOCAB + 2
        self.sd = synth.make_state_dict(C, seed=0)
        if backend == "engine":
            self.model = hctr_amd.hctr_model(C, precision=precision).cuda(0)
            self.model.load_state_dict(self.sd)

    def __call__(self, imgs):
        """[n, W, 2048] float32 head inputs in the reference's feature order d = c*4 + h"""
        n = imgs.shape[0]
        if self.backend == "engine":
            self.model.greedy(imgs)
            act = self.model.debug_activation("stage4", n)             # [n, 512, 4, W] of the LAST internal pass
            if act.size != n * 2048 * imgs.shape[2]:
                raise ValueError("batch of %d lines was split into internal passes: lower --batch" % n)
        else:
            raise ValueError("only the engine backend exists (the oracle is test infrastructure)")
        return np.ascontiguousarray(act.transpose(0, 3, 1, 2).reshape(n, imgs.shape[2], 2048))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="engine", choices=["engine"])
    ap.add_argument("--train-lines", type=int, default=1536)
    ap.add_argument("--eval-lines", type=int, default=64)
    ap.add_argument("--width", type=int, default=2000)
    ap.add_argument("--classes", type=int, default=synth.FONT_CLASSES)
    ap.add_argument("--batch", type=int, default=16,
                    help="lines per feature pass (engine backend: at most one internal pass, 21 lines of width 2000 in f16x3)")
    ap.add_argument("--lam", type=float, default=1e-3, help="ridge, relative to mean diag of the Gram matrix")
    ap.add_argument("--class-weight", type=float, default=0.0, help="weight of class columns (0 = balance with blank)")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(synth.__file__), "synth_head_trained.npz"))
    args = ap.parse_args()
    K, W = args.classes, args.width
    feats = Features(args.backend)
    D = 2048
    # pass 1: weighted moments. sample weight: 1 for blank columns, cw for class columns
    S1 = np.zeros((D,), np.float64)
    Sxx = np.zeros((D, D), np.float64)
    Sxy = np.zeros((D, K + 1), np.float64)
    Sy = np.zeros((K + 1,), np.float64)
    sw = 0.0
    cw = args.class_weight
    t0 = time.time()
    for s in range(0, args.train_lines, args.batch):
        n = min(args.batch, args.train_lines - s)
        imgs, truth = synth.make_font_lines(n, W, TRAIN_SEED, line_offset=s, with_truth=True, n_classes=K)
        X = feats(imgs).reshape(n * W, D)
        T = np.concatenate([column_targets(b, W) for b in truth])
        keep = T >= 0
        X, T = X[keep], T[keep]
        if cw <= 0:
            cw = float((T == 0).sum()) / max(1, int((T > 0).sum()))        # balance (fixed from the first batch)
            print("class weight", cw, flush=True)
        wgt = np.where(T > 0, cw, 1.0).astype(np.float32)
        Xw = X * wgt[:, None]
        Sxx += (Xw.T @ X).astype(np.float64)
        S1 += Xw.sum(axis=0, dtype=np.float64)
        sw += float(wgt.sum(dtype=np.float64))
        order = np.argsort(T, kind="stable")
        bounds = np.searchsorted(T[order], np.arange(K + 2))
        for k in range(K + 1):
            idx = order[bounds[k]:bounds[k + 1]]
            if idx.size:
                Sxy[:, k] += Xw[idx].sum(axis=0, dtype=np.float64)
                Sy[k] += float(wgt[idx].sum(dtype=np.float64))
        print("train lines %d..%d  %.0f s" % (s, s + n - 1, time.time() - t0), flush=True)
    mu = S1 / sw
    ybar = Sy / sw
    Cxx = Sxx / sw - np.outer(mu, mu)
    Cxy = Sxy / sw - np.outer(mu, ybar)
    lam = args.lam * float(np.trace(Cxx)) / D
    Wf = np.linalg.solve(Cxx + lam * np.eye(D), Cxy)                   # [D, K+1]
    bf = ybar - mu @ Wf
    w16 = (ALPHA * Wf.T).astype(np.float16)                            # stored rows; the checkpoint uses float32(w16)
    b32 = (ALPHA * bf).astype(np.float32)
    labels = np.array([0] + [synth.font_label(k) for k in range(K)], np.int32)

    # held-out check on the bench's lines (seed 2), with the weights exactly as stored; with the engine backend also
    # the same read-out on the f16-mode features: how many columns / lines would the default precision decode differently
    imgs, truth = synth.make_font_lines(args.eval_lines, W, 2, with_truth=True, n_classes=K)
    chars = synth.characters()
    Wev = w16.astype(np.float32)
    ok = edits = nchar = 0
    margins = []
    import bench
    f16 = Features("engine", "f16") if args.backend == "engine" else None
    flips = lines_diff = runs = 0
    max_dlogit = 0.0

    def decode(a):
        keep = (a != 0) & np.concatenate([[True], a[1:] != a[:-1]])
        return "".join(chars[labels[c] - 1] for c in a[keep])

    for s in range(0, args.eval_lines, args.batch):
        X = feats(imgs[s:s + args.batch])
        lg = X @ Wev.T + b32                                            # [n, W, K+1]
        srt = np.sort(lg, axis=2)
        margins.append((srt[:, :, -1] - srt[:, :, -2]).ravel())
        am = lg.argmax(axis=2)
        if f16 is not None:
            lg16 = f16(imgs[s:s + args.batch]) @ Wev.T + b32
            am16 = lg16.argmax(axis=2)
            flips += int((am16 != am).sum())
            max_dlogit = max(max_dlogit, float(np.abs(lg16 - lg).max()))
        for i in range(am.shape[0]):
            text = decode(am[i])
            want = synth.font_truth_text(truth[s + i], W, chars)
            ok += text == want
            edits += bench.edit_distance(text, want)
            nchar += len(want)
            runs += len(text)
            if f16 is not None:
                lines_diff += decode(am16[i]) != text
    m = np.concatenate(margins)
    edges = [0, 0.01, 0.03, 0.1, 0.3, 1, 3, 10, 1e9]
    rep = {"train_lines": args.train_lines, "classes": K, "lam": args.lam, "class_weight": cw, "alpha": ALPHA,
           "eval_lines": args.eval_lines, "eval_exact_lines_vs_truth": int(ok), "eval_cer_vs_truth": edits / max(1, nchar),
           "eval_margin_hist": {"edges": edges[:-1], "counts": np.histogram(m, bins=edges)[0].tolist()},
           "eval_margin_min": float(m.min()), "eval_columns": int(m.size), "eval_decoded_chars": runs,
           "eval_true_chars": nchar, "logit_scale": float(np.abs(lg).max()),
           "f16_vs_f16x3": None if f16 is None else {"argmax_flips": flips, "lines_with_different_text": int(lines_diff),
                                                      "max_abs_dlogit": max_dlogit}}
    print(json.dumps(rep))
    np.savez_compressed(args.out, w=w16, b=b32)        # row 0 = <blank>, row 1+k = font class k (synth.font_label(k))
    with open(os.path.splitext(args.out)[0] + ".json", "w") as f:
        json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
