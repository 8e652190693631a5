"""Config-2 parity report against the REAL reference's fixtures for all 64 lines (run on the GPU box):

    python tools/gpu_c2_parity.py [--checkpoint random|trained] [--modes f16,f16x3]

Per precision mode: error of the per-column max logit and log-sum-exp vs the fp32 CPU reference, argmax flips
bucketed by the reference's own top-2 margin, the largest margin among flipped columns, exact lines and
character edits of the greedy text. Prints one JSON object; tests/test_gpu_parity.py asserts fixed floors
derived from these numbers.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import hctr_amd  # noqa: E402


def report(model, codec, imgs, meta, gold, chunk=8):
    n, W = imgs.shape[0], imgs.shape[2]
    amax = np.zeros((n, W), np.int64)
    mx = np.zeros((n, W), np.float32)
    lse = np.zeros((n, W), np.float32)
    for s in range(0, n, chunk):
        lg = model(imgs[s:s + chunk])                     # [W, b, C] float32 on the host
        amax[s:s + chunk] = lg.argmax(axis=2).T
        m = lg.max(axis=2)
        mx[s:s + chunk] = m.T
        lse[s:s + chunk] = (m.astype(np.float64) + np.log(np.exp(lg.astype(np.float64) - m[..., None]).sum(axis=2))).T
    text = codec.labels_to_text(model.greedy(imgs))
    ref_arg = gold["argmax"][:n].astype(np.int64)
    margin = gold["margin"][:n]
    flips = amax != ref_arg
    # a flip to the reference's own runner-up is the benign kind (top-2 swapped)
    to_second = flips & (amax == gold["second"][:n].astype(np.int64))
    edges = [0, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 3e-2, 0.1, 0.3, 1.0, 1e9]
    out = {"lines": n, "columns": int(flips.size), "logit_scale": float(np.abs(gold["max"][:n]).max()),
           "max_err_of_column_max": float(np.abs(mx - gold["max"][:n]).max()),
           "p999_err_of_column_max": float(np.quantile(np.abs(mx - gold["max"][:n]), 0.999)),
           "max_err_of_lse": float(np.abs(lse - gold["lse"][:n]).max()),
           "argmax_flips": int(flips.sum()), "flips_to_reference_runner_up": int(to_second.sum()),
           "largest_margin_among_flips": float(margin[flips].max()) if flips.any() else 0.0,
           "flips_by_margin": {"edges": edges[:-1], "flips": np.histogram(margin[flips], bins=edges)[0].tolist(),
                               "columns": np.histogram(margin, bins=edges)[0].tolist()}}
    out.update(bench.text_parity(text, meta, n))
    out["lines_with_a_flip"] = int(flips.any(axis=1).sum())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", default="random", choices=["random", "trained"])
    ap.add_argument("--modes", default="f16,f16x3")
    ap.add_argument("--lines", type=int, default=64)
    args = ap.parse_args()
    synth = hctr_amd.synth
    C = synth.DEFAULT_VOCAB + 2
    sd = bench.make_checkpoint(synth, C, args.checkpoint)
    meta, gold = bench.load_c2_golden() if args.checkpoint == "random" else bench.load_c2_golden_trained()
    imgs = bench.make_lines(synth, args.checkpoint, args.lines, bench.W_LINE, bench.SEED, 0)
    codec = hctr_amd.ctc_codec(synth.characters())
    res = {"checkpoint": args.checkpoint}
    for mode in args.modes.split(","):
        m = hctr_amd.hctr_model(C, precision=mode).cuda(0)
        m.load_state_dict(sd)
        res[mode] = report(m, codec, imgs, meta, gold)
        del m
    print(json.dumps(res))


if __name__ == "__main__":
    main()
