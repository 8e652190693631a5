"""Where does the f16 mode's logit error come from?  (diagnostic, run on the GPU box:  python tests/diag_precision_attribution.py)

The f16x3 machinery carries every activation and weight as a hi + lo fp16 pair. HCTR_X3_MASK (engine.cpp) lets ONE class
of rounding points at a time keep its lo part while every other class is rounded to a single fp16 value exactly as the
f16 mode does:
    1  conv weights            2  conv1 outputs of the blocks      4  block outputs (the residual stream)
    8  stem + stage-conv outputs                                   16  head input and head weights
mask 0 is therefore the f16 arithmetic (run through the 3x kernels), mask 31 the f16x3 mode. For every mask the 64 lines
of BASELINE configs[1] (random-head checkpoint, the near-tie-rich one) are compared with the REAL reference's outputs
(tests/golden/c2_lines.*, fp32 CPU): largest error of a column's maximum logit, argmax flips of 128 000 columns, lines
with exactly the reference's text, character edits - next to the MFMA work a mode that carried only those classes would
cost (products per term, FLOP-weighted over the layers whose operands carry a lo part).

Writes profiles/r03_precision_attribution.json (and prints a markdown table for DESIGN.md section 4).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hctr_amd  # noqa: E402

synth = hctr_amd.synth
C = synth.DEFAULT_VOCAB + 2
GOLD = os.path.join(ROOT, "tests", "golden")


def edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def cost_factor(mask):
    """MFMA products per term of a mode that carries lo parts only for the classes in `mask`, FLOP-weighted: a conv costs
    1 + [its weights carry lo] + [its input activation carries lo] products (MFLOP per column from SURVEY 8a)."""
    w = 1 if mask & 1 else 0
    layers = []                                   # (MFLOP per column, input class bit, is_head)
    layers.append((9.44, 8, False))               # conv0_2 reads the stem output
    # stage s: block i conv1 reads the stage input (class 8) or the previous block's output (4); conv2 reads conv1's (2)
    plan = [(2, 9.44, 18.87, 18.87), (4, 18.87, 37.75, 37.75), (5, 37.75, 75.50, 75.50), (1, 37.75, 37.75, 37.75)]
    for nblk, first_c1, other_c1, c2 in plan:
        for i in range(nblk):
            layers.append((first_c1 if i == 0 else other_c1, 8 if i == 0 else 4, False))
            layers.append((c2, 2, False))
        layers.append((c2 if nblk != 1 else 37.75, 4, False))       # the stage conv reads the last block's output
    layers.append((30.14, 16, True))
    tot = sum(f for f, _, _ in layers)
    cost = 0.0
    for f, cls, head in layers:
        wl = (1 if mask & 16 else 0) if head else w
        cost += f * (1 + wl + (1 if mask & cls else 0))
    return cost / tot


def main():
    with open(os.path.join(GOLD, "c2_lines.json"), encoding="utf-8") as f:
        meta = json.load(f)
    g = np.load(os.path.join(GOLD, "c2_lines.npz"))
    ref_arg, ref_max, margin = g["argmax"].astype(np.int64), g["max"], g["margin"]
    scale = float(np.abs(ref_max).max())
    imgs = synth.make_line_images(64, 2000, meta["seed"])
    sd = synth.make_state_dict(C, seed=0)
    cd = hctr_amd.ctc_codec(synth.characters())
    names = {0: "none (= f16 arithmetic)", 1: "conv weights", 2: "conv1 outputs", 4: "block outputs (residual stream)",
             8: "stem + stage-conv outputs", 16: "head input + head weights", 6: "all block activations (2+4)",
             14: "all trunk activations (2+4+8)", 30: "all activations + head (no conv-weight lo)",
             15: "whole trunk (weights + activations), f16 head", 31: "everything (= f16x3)"}
    rows = []
    for mask in (0, 1, 2, 4, 8, 16, 6, 14, 30, 15, 31):
        os.environ["HCTR_X3_MASK"] = str(mask)
        m = hctr_amd.hctr_model(C, precision="f16x3").cuda(0)
        m.load_state_dict(sd)
        arg = np.zeros_like(ref_arg)
        mx = np.zeros_like(ref_max)
        for s0 in range(0, 64, 8):
            lg = m(imgs[s0:s0 + 8])
            arg[s0:s0 + 8] = lg.argmax(axis=2).T
            mx[s0:s0 + 8] = lg.max(axis=2).T
        text = cd.labels_to_text(m.greedy(imgs))
        flips = arg != ref_arg
        row = {"mask": mask, "classes_with_lo": names[mask], "max_err_of_column_max": round(float(np.abs(mx - ref_max).max()), 5),
               "rms_err_of_column_max": round(float(np.sqrt(np.mean((mx - ref_max) ** 2))), 5),
               "argmax_flips_of_128000": int(flips.sum()),
               "largest_reference_margin_among_flips": round(float(margin[flips].max()), 5) if flips.any() else 0.0,
               "lines_exact_of_64": int(sum(a == b for a, b in zip(text, meta["greedy"]))),
               "char_edits_of_%d" % sum(len(t) for t in meta["greedy"]): int(sum(edit_distance(a, b) for a, b in zip(text, meta["greedy"]))),
               "mfma_cost_factor": round(cost_factor(mask), 2)}
        rows.append(row)
        print(json.dumps(row), flush=True)
        del m
    os.environ.pop("HCTR_X3_MASK", None)
    out = {"workload": "BASELINE configs[1], 64 x 1x128x2000, random-head checkpoint, vs tests/golden/c2_lines.* (REAL reference, fp32 CPU)",
           "logit_scale": scale, "rows": rows}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for path in (os.path.join(ROOT, "gpurun_out", "r03_precision_attribution.json"),):
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
    print("\n| lo parts kept for | max err of column max | rms | flips / 128 000 | largest ref. margin flipped | lines exact / 64 | edits | MFMA cost |")
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        ed = [v for k, v in r.items() if k.startswith("char_edits")][0]
        print("| %s | %.4f | %.4f | %d | %.4f | %d | %d | %.2fx |" % (r["classes_with_lo"], r["max_err_of_column_max"],
              r["rms_err_of_column_max"], r["argmax_flips_of_128000"], r["largest_reference_margin_among_flips"],
              r["lines_exact_of_64"], ed, r["mfma_cost_factor"]))


if __name__ == "__main__":
    main()
