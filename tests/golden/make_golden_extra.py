"""Additional golden fixtures from the REAL reference (same rules as make_golden.py: build container only,
inputs by seed, expected outputs only - no reference source copied).

Adds batches with strongly unequal widths (the pad region dominates some lines' SE means) and a longer two-line
batch with beam-search strings:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_extra.py
        -> tests/golden/model_extra.npz, tests/golden/model_extra_strings.json
"""
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (sets up sys.path for the reference and the package)

import torch  # noqa: E402
from models.handwritten_ctr_model import hctr_model  # noqa: E402  (reference)
from utils.ctc_codec import ctc_codec  # noqa: E402             (reference)

synth, hctr_ref, ctc_ref, codec_cases = mg.synth, mg.hctr_ref, mg.ctc_ref, mg.codec_cases
CASES = [("b2w300u", 51, [300, 211]), ("b4w131u", 52, [131, 100, 64, 17])]


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    sd = synth.make_state_dict(mg.C, seed=0)
    model = hctr_model(mg.C)
    model.load_state_dict(synth.to_torch(sd), strict=True)
    model.eval()
    chars = synth.characters()
    codec = ctc_codec(chars)
    out = {"sub_classes": mg.SUB_CLASSES}
    strings = {}
    for name, seed, widths in CASES:
        imgs = synth.make_line_images(len(widths), max(widths), seed)
        x = synth.normalize_pad(imgs, widths)
        taps = {}
        with torch.no_grad():
            ref = model(torch.from_numpy(x)).numpy()
        mine = hctr_ref.forward(sd, x, taps).numpy()
        assert np.abs(ref - mine).max() <= 1e-4, "oracle restatement drifted from the reference"
        out[name + "/logits_sub"] = ref[:, :, mg.SUB_CLASSES].astype(np.float32)
        for k, v in mg.column_stats(ref).items():
            out[name + "/" + k] = v
        for k in ("stage0", "stage1", "stage2", "stage3", "stage4", "block1.0", "block3.4"):
            out[name + "/act/" + k] = taps[k][:, :8, :, :16].numpy().astype(np.float32)
        entry = {"widths": widths, "seed": seed, "greedy": codec.decode(ref)}
        if name == "b2w300u":
            for tag, skip, lm, lp, lb, bs, depth in codec_cases.BEAM_SETTINGS[:4]:
                cdc = ctc_codec(chars)
                cdc.use_beam_search, cdc.skip_search = True, skip
                cdc.use_tfm_pred, cdc.use_tfm_score = False, False
                cdc.lm_panelty, cdc.len_bonus, cdc.beam_size, cdc.search_depth = lp, lb, bs, depth
                cdc.ngram = ctc_ref.ZeroLM() if lm == "zero" else ctc_ref.ToyBigramLM()
                entry[tag] = cdc.decode(ref)
        strings[name] = entry
        print(name, "greedy lens", [len(s) for s in entry["greedy"]])
    np.savez_compressed(os.path.join(HERE, "model_extra.npz"), **out)
    with open(os.path.join(HERE, "model_extra_strings.json"), "w") as f:
        json.dump(strings, f, ensure_ascii=False, indent=1)


if __name__ == "__main__":
    main()
