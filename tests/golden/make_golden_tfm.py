"""Golden strings for the transformer-LM hooks from the REAL reference codec.

The reference's ``ctc_codec`` duck-types its transformer (utils/ctc_codec.py:215-227,269-274), so the real
class runs here with ``codec_cases.FakeTransformer`` attached in place of the fairseq model nobody ships.
Stores the decoded strings only (tests/golden/codec_tfm.json). Build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_tfm.py
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")

import codec_cases  # noqa: E402
from oracle import ctc_ref  # noqa: E402  (toy n-gram object only)
from utils.ctc_codec import ctc_codec  # noqa: E402  (reference)


def main():
    out = {}
    for name, seed, w, b, c, style, use_score, use_pred, ragged in codec_cases.TFM_CASES:
        chars = codec_cases.vocab(c)
        logits = codec_cases.gen_logits(seed, w, b, c, style)
        cdc = ctc_codec(chars)
        cdc.use_beam_search, cdc.skip_search = True, False
        cdc.use_tfm_score, cdc.use_tfm_pred = use_score, use_pred
        for k, v in codec_cases.TFM_SETTINGS.items():
            setattr(cdc, k, v)
        cdc.transformer = codec_cases.FakeTransformer(cdc.characters[1:-1], ragged)
        cdc.ngram = ctc_ref.ToyBigramLM()
        try:
            out[name] = cdc.decode(logits)
        except IndexError:
            out[name] = "IndexError"
        print(name, out[name] if isinstance(out[name], str) else [len(s) for s in out[name]])
    with open(os.path.join(HERE, "codec_tfm.json"), "w") as f:
        json.dump(out, f, ensure_ascii=False, indent=1)


if __name__ == "__main__":
    main()
