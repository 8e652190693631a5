"""CPU tests: the oracle (oracle/) is pinned to outputs of the REAL reference.

The fixtures under tests/golden/ were produced by tests/golden/make_golden.py, which imports the
reference's hctr_model / ctc_codec in the build container. These tests re-run the oracle's
restatement on the same seeded inputs and require agreement: float tolerance for logits
(CPU conv kernels may differ in summation order between hosts), exact for indices/strings.
"""
import json
import os
import zlib

import numpy as np
import pytest

import codec_cases
from conftest import GOLDEN
from oracle import ctc_ref, hctr_ref

LOGIT_ATOL = 2e-4     # fp32 CPU-vs-CPU (different thread counts / ISAs); logits are O(10)


def _strings():
    with open(os.path.join(GOLDEN, "model_strings.json")) as f:
        return json.load(f)


def test_synth_known_answers(synth, state_dict):
    with open(os.path.join(GOLDEN, "synth_kat.json")) as f:
        kat = json.load(f)
    assert len(state_dict) == 254 and set(kat) == set(state_dict)
    for k, v in state_dict.items():
        a = np.ascontiguousarray(v)
        assert list(a.shape) == kat[k]["shape"] and str(a.dtype) == kat[k]["dtype"], k
        if k == "linear.bias":       # contains a float64 mat-vec: allow 1-ulp BLAS differences
            np.testing.assert_allclose(a.reshape(-1)[:4], kat[k]["head"], rtol=1e-6)
            continue
        assert (zlib.crc32(a.tobytes()) & 0xFFFFFFFF) == kat[k]["crc32"], k


@pytest.mark.parametrize("name,seed,widths", [("b1w32", 21, [32]), ("b3w67u", 22, [67, 50, 33]),
                                              ("b2w96", 23, [96, 96])])
def test_oracle_forward_matches_reference(synth, state_dict, name, seed, widths):
    g = np.load(os.path.join(GOLDEN, "model_small.npz"))
    imgs = synth.make_line_images(len(widths), max(widths), seed)
    x = synth.normalize_pad(imgs, widths)
    taps = {}
    logits = hctr_ref.forward(state_dict, x, taps).numpy()
    assert logits.shape == (max(widths), len(widths), synth.DEFAULT_VOCAB + 2)
    sub = g["sub_classes"]
    np.testing.assert_allclose(logits[:, :, sub], g[name + "/logits_sub"], atol=LOGIT_ATOL, rtol=0)
    np.testing.assert_allclose(logits.max(axis=2), g[name + "/max"], atol=LOGIT_ATOL, rtol=0)
    for k in ("stage0", "stage2", "stage4", "block3.4"):
        np.testing.assert_allclose(taps[k][:, :8, :, :16].numpy(), g[name + "/act/" + k], atol=1e-4, rtol=0)
    # indices: exact wherever the reference's own top-2 margin exceeds the float tolerance
    margin = g[name + "/top10_val"][:, :, 0] - g[name + "/top10_val"][:, :, 1]
    safe = margin > 4 * LOGIT_ATOL
    assert safe.mean() > 0.99
    assert np.array_equal(logits.argmax(axis=2)[safe], g[name + "/argmax"][safe].astype(np.int64))
    codec = ctc_ref.CtcCodecRef(synth.characters())
    if safe.all():
        assert codec.decode(logits) == _strings()[name]["greedy"]


def test_codec_cases_match_reference():
    with open(os.path.join(GOLDEN, "codec_cases.json")) as f:
        gold = json.load(f)
    for name, seed, w, b, c, style in codec_cases.CODEC_CASES:
        logits = codec_cases.gen_logits(seed, w, b, c, style)
        chars = codec_cases.vocab(c)
        codec = ctc_ref.CtcCodecRef(chars)
        assert codec.decode(logits) == gold[name]["greedy"], name
        enc = codec.encode(["".join(chars[:3]) + "?", "", chars[-1]])
        assert [enc[0].tolist(), enc[1].tolist()] == gold[name]["encode"]
        for tag, skip, lm, lp, lb, bs, depth in codec_cases.BEAM_SETTINGS:
            codec = ctc_ref.CtcCodecRef(chars)
            codec.use_beam_search, codec.skip_search = True, skip
            codec.use_tfm_pred = False
            codec.lm_panelty, codec.len_bonus, codec.beam_size, codec.search_depth = lp, lb, bs, depth
            codec.ngram = ctc_ref.ZeroLM() if lm == "zero" else ctc_ref.ToyBigramLM()
            try:
                got = codec.decode(logits)
            except IndexError:
                got = "IndexError"
            assert got == gold[name][tag], (name, tag)


def test_codec_greedy_rules():
    """Collapse rules of utils/ctc_codec.py:89-93, spelled out."""
    codec = ctc_ref.CtcCodecRef("AB")            # C = 4: blank, A, B, unknown
    def lg(seq):
        x = np.zeros((len(seq), 1, 4), np.float32)
        for t, s in enumerate(seq):
            x[t, 0, s] = 1.0
        return x
    assert codec.decode(lg([1, 1, 2])) == ["AB"]
    assert codec.decode(lg([1, 0, 1])) == ["AA"]          # blank separates repeats
    assert codec.decode(lg([1, 3, 1])) == ["AA"]          # unknown is dropped but still separates
    assert codec.decode(lg([0, 0, 3])) == [""]
    assert codec.decode(np.zeros((3, 1, 4), np.float32)) == [""]   # ties -> first index (blank)
    assert codec.decode(np.zeros((0, 2, 4), np.float32)) == []     # zero-length lines are skipped


def test_edit_distance():
    assert ctc_ref.edit_distance("kitten", "sitting") == 3
    assert ctc_ref.edit_distance("", "abc") == 3
    assert ctc_ref.edit_distance("abc", "abc") == 0


@pytest.mark.parametrize("name,seed,widths", [("b2w300u", 51, [300, 211]), ("b4w131u", 52, [131, 100, 64, 17])])
def test_oracle_matches_reference_extra_fixtures(synth, state_dict, name, seed, widths):
    """tests/golden/model_extra.npz (make_golden_extra.py: the REAL reference on strongly unequal widths, where
    the replicate pad dominates some lines' SE means), incl. its beam-search strings on the same logits."""
    g = np.load(os.path.join(GOLDEN, "model_extra.npz"))
    with open(os.path.join(GOLDEN, "model_extra_strings.json"), encoding="utf-8") as f:
        strings = json.load(f)[name]
    imgs = synth.make_line_images(len(widths), max(widths), seed)
    taps = {}
    logits = hctr_ref.forward(state_dict, synth.normalize_pad(imgs, widths), taps).numpy()
    sub = g["sub_classes"]
    np.testing.assert_allclose(logits[:, :, sub], g[name + "/logits_sub"], atol=LOGIT_ATOL, rtol=0)
    np.testing.assert_allclose(logits.max(axis=2), g[name + "/max"], atol=LOGIT_ATOL, rtol=0)
    for k in ("stage1", "stage3", "block1.0", "block3.4"):
        np.testing.assert_allclose(taps[k][:, :8, :, :16].numpy(), g[name + "/act/" + k], atol=1e-4, rtol=0)
    margin = g[name + "/top10_val"][:, :, 0] - g[name + "/top10_val"][:, :, 1]
    safe = margin > 4 * LOGIT_ATOL
    assert np.array_equal(logits.argmax(axis=2)[safe], g[name + "/argmax"][safe].astype(np.int64))
    codec = ctc_ref.CtcCodecRef(synth.characters())
    if safe.all():
        assert codec.decode(logits) == strings["greedy"]
        for tag, skip, lm, lp, lb, bs, depth in codec_cases.BEAM_SETTINGS[:4]:
            if tag not in strings:
                continue
            oc = ctc_ref.CtcCodecRef(synth.characters())
            oc.use_beam_search, oc.skip_search, oc.use_tfm_pred = True, skip, False
            oc.lm_panelty, oc.len_bonus, oc.beam_size, oc.search_depth = lp, lb, bs, depth
            oc.ngram = ctc_ref.ZeroLM() if lm == "zero" else ctc_ref.ToyBigramLM()
            assert oc.decode(logits) == strings[tag], tag


@pytest.mark.parametrize("which,line", [("random", 5), ("trained", 11)])
def test_oracle_matches_reference_on_config2_lines(synth, state_dict, which, line):
    """One full-width (W=2000) line of BASELINE configs[1] per checkpoint: the oracle forward + codec against the REAL
    reference's per-column outputs (tests/golden/c2_lines.* / c2_trained_lines.*, all 64 lines; one is re-run here)."""
    base = "c2_lines" if which == "random" else "c2_trained_lines"
    with open(os.path.join(GOLDEN, base + ".json"), encoding="utf-8") as f:
        meta = json.load(f)
    g = np.load(os.path.join(GOLDEN, base + ".npz"))
    C = synth.DEFAULT_VOCAB + 2
    if which == "random":
        sd, imgs = state_dict, synth.make_line_images(1, meta["width"], meta["seed"], line_offset=line)
    else:
        sd = synth.make_state_dict(C, seed=0, head="trained")
        imgs = synth.make_font_lines(1, meta["width"], meta["seed"], line_offset=line)
    logits = hctr_ref.forward(sd, synth.normalize_pad(imgs)).numpy()[:, 0, :]
    np.testing.assert_allclose(logits.max(axis=1), g["max"][line], atol=3 * LOGIT_ATOL, rtol=0)
    safe = g["margin"][line] > 4 * LOGIT_ATOL
    assert safe.mean() > 0.99
    assert np.array_equal(logits.argmax(axis=1)[safe], g["argmax"][line][safe].astype(np.int64))
    if safe.all():
        assert ctc_ref.CtcCodecRef(synth.characters()).decode(logits[:, None, :])[0] == meta["greedy"][line]


def test_trained_checkpoint_fixture_is_peaky(synth):
    """The trained-like checkpoint's reason to exist, stated on the REAL reference's outputs: top-2 margins far above any
    reduced-precision logit error (f16: <= 1 % of the logit scale), and the reference reads the font lines nearly
    right (its greedy text vs the lines' ground truth)."""
    with open(os.path.join(GOLDEN, "c2_trained_lines.json"), encoding="utf-8") as f:
        meta = json.load(f)
    g = np.load(os.path.join(GOLDEN, "c2_trained_lines.npz"))
    scale = float(np.abs(g["max"]).max())
    margin = g["margin"]
    assert (margin < 0.02 * scale).sum() <= 40                       # of 128 000 columns (random head: ~17 000)
    assert margin.min() > 0.002 * scale
    _, truth = synth.make_font_lines(64, meta["width"], meta["seed"], with_truth=True)
    edits = sum(ctc_ref.edit_distance(t, synth.font_truth_text(b, meta["width"])) for t, b in zip(meta["greedy"], truth))
    assert edits <= 0.05 * sum(len(t) for t in meta["greedy"])


def test_fast_toy_checker_equals_the_plain_oracle_and_the_reference():
    """bench.py's config-5 checker (oracle.ctc_ref.FastToyCodecRef on the device's top-k: memoised toy-LM prefix sums,
    search fed with the top-k classes instead of the full log-prob rows) must be the SAME search: on every codec case it
    reproduces the strings the REAL reference codec produced with the toy-bigram LM (tests/golden/codec_cases.json)."""
    from scipy.special import log_softmax
    with open(os.path.join(GOLDEN, "codec_cases.json"), encoding="utf-8") as f:
        gold = json.load(f)
    checked = 0
    for name, seed, w, b, c, style in codec_cases.CODEC_CASES:
        logits = codec_cases.gen_logits(seed, w, b, c, style)
        lp = log_softmax(logits, axis=2)
        for tag, skip, lm, pen, bonus, beam, depth in codec_cases.BEAM_SETTINGS:
            if skip or lm != "toy":
                continue
            oc = ctc_ref.FastToyCodecRef(codec_cases.vocab(c))
            oc.use_beam_search, oc.use_tfm_pred, oc.skip_search = True, False, False
            oc.lm_panelty, oc.len_bonus, oc.beam_size, oc.search_depth = pen, bonus, beam, depth
            k = min(depth, c)
            oc.search_depth = k
            topk = np.flip(np.argsort(lp, axis=2), axis=2)[:, :, :k]
            tlp = np.take_along_axis(lp, topk, axis=2).astype(np.float32)
            try:
                got = oc.beam_full_from_topk(topk, tlp)
            except IndexError:
                got = "IndexError"
            assert got == gold[name][tag], (name, tag)
            checked += 1
    assert checked >= 10
