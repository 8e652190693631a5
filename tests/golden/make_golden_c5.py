"""Golden strings for BASELINE configs[4] (beam search 10/10 on full-width lines) from the REAL reference, end to end:
the reference's ``hctr_model`` forward (fp32 CPU) on trained-like-checkpoint font lines of width 2000, then the
reference's ``ctc_codec`` beam decode (cbs_full and cbs_skip, toy-bigram and zero LM, test.py:74-79 hyper-parameters).
Stores the strings only (tests/golden/c5_beam_lines.json). Build container only, ~1 minute.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_c5.py
"""
import importlib
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402

synth = importlib.import_module("handwritten-chinese-ocr-samples_amd.synth")
from oracle import ctc_ref  # noqa: E402  (toy LM objects only)
from models.handwritten_ctr_model import hctr_model  # noqa: E402  (reference)
from utils.ctc_codec import ctc_codec  # noqa: E402             (reference)

LINES, W, SEED = 6, 2000, 5
SETTINGS = [("full_toy", False, "toy", 0.8, 4.8), ("full_zero", False, "zero", 0.8, 4.8), ("skip_toy", True, "toy", 0.8, 4.8)]


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    C = synth.DEFAULT_VOCAB + 2
    sd = synth.make_state_dict(C, seed=0, head="trained")
    model = hctr_model(C)
    model.load_state_dict(synth.to_torch(sd), strict=True)
    model.eval()
    imgs = synth.make_font_lines(LINES, W, SEED)
    out = {"lines": LINES, "width": W, "seed": SEED, "beam_size": 10, "search_depth": 10}
    for s in range(0, LINES, 3):
        with torch.no_grad():
            logits = model(torch.from_numpy(synth.normalize_pad(imgs[s:s + 3]))).numpy()
        cdc = ctc_codec(synth.characters())
        out.setdefault("greedy", []).extend(cdc.decode(logits))
        for tag, skip, lm, lp, lb in SETTINGS:
            cdc = ctc_codec(synth.characters())
            cdc.use_beam_search, cdc.skip_search = True, skip
            cdc.use_tfm_pred, cdc.use_tfm_score = False, False
            cdc.lm_panelty, cdc.len_bonus, cdc.beam_size, cdc.search_depth = lp, lb, 10, 10
            cdc.ngram = ctc_ref.ZeroLM() if lm == "zero" else ctc_ref.ToyBigramLM()
            out.setdefault(tag, []).extend(cdc.decode(logits))
        print("lines", s, "done", flush=True)
    with open(os.path.join(HERE, "c5_beam_lines.json"), "w") as f:
        json.dump(out, f, ensure_ascii=False, indent=1)
    for tag, *_ in SETTINGS:
        print(tag, [len(t) for t in out[tag]], "differs from greedy:", sum(a != b for a, b in zip(out[tag], out["greedy"])))


if __name__ == "__main__":
    main()
