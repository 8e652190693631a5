"""GPU tests (-m gpu) of the guarded precision mode (``precision="auto"``, C ABI ``hctr_set_precision(ctx, 2)``):
every line runs in f16; the fused head also yields each column's top-1/top-2 logit margin; a line with a column whose
margin is within twice the f16 logit tolerance (LOGIT_RTOL * max|logit of the line| + LOGIT_ATOL - the tolerance
tests/test_gpu_parity.py asserts for f16) is run again in f16x3 at the same padded width.

What must hold:
  * the text of auto mode EQUALS the text of f16x3 mode on every fixture (flagged lines ARE f16x3 results; an unflagged
    line's f16 argmax cannot differ from an fp32-grade one within the asserted tolerance);
  * on the trained-like checkpoint, the ragged config-3 batch and the config-5 beam lines it equals the REAL reference's
    strings (tests/golden/*.json, written by the reference itself);
  * the guard figures are exactly the margins / magnitudes of the engine's own f16 logits;
  * logits and beam front-end outputs of flagged lines are the f16x3 mode's, bit for bit, the others the f16 mode's.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

LOGIT_RTOL, LOGIT_ATOL = 0.01, 0.05          # = tests/test_gpu_parity.py (and the engine's guard defaults)


@pytest.fixture(scope="module")
def auto_random(pkg, synth, state_dict):
    """one context holding both weight sets of the random-head checkpoint: serves f16, f16x3 and auto"""
    m = pkg.hctr_model(synth.DEFAULT_VOCAB + 2, precision="auto").cuda(0)
    m.load_state_dict(state_dict)
    return m


@pytest.fixture(scope="module")
def auto_trained(pkg, synth):
    C = synth.DEFAULT_VOCAB + 2
    m = pkg.hctr_model(C, precision="auto").cuda(0)
    m.load_state_dict(synth.make_state_dict(C, seed=0, head="trained"))
    return m


def _texts(pkg, synth, m, imgs, widths=None):
    """greedy text in the three modes of one context + the guard figures of the auto call"""
    cd = pkg.ctc_codec(synth.characters())
    out = {}
    for mode in ("f16", "f16x3", "auto"):
        m.set_precision(mode)
        out[mode] = cd.labels_to_text(m.greedy(imgs, widths=widths))
        if mode == "auto":
            out["guard"] = m.last_guard()
    return out


def _np_guard(logits):
    """[W,B,C] logits -> per line (min over columns of top1 - top2, max |logit|)"""
    srt = np.sort(logits, axis=2)
    margin = srt[:, :, -1] - srt[:, :, -2]
    return margin.min(axis=0), np.abs(logits).max(axis=(0, 2))


@pytest.mark.parametrize("seed,widths", [(21, [32]), (22, [67, 50, 33]), (23, [96, 96]), (51, [300, 211]),
                                         (52, [131, 100, 64, 17]), (31, [488])])
def test_auto_equals_f16x3_and_guard_figures_are_the_f16_margins(pkg, synth, auto_random, seed, widths):
    m = auto_random
    imgs = synth.make_line_images(len(widths), max(widths), seed)
    wd = np.array(widths, np.int32)
    t = _texts(pkg, synth, m, imgs, wd)
    assert t["auto"] == t["f16x3"]
    g = t["guard"]
    assert g["lines"] == len(widths) and g["flagged"] == int(g["flags"].sum())
    # the figures are those of the engine's own f16 logits (same head GEMM, same accumulators): exact
    m.set_precision("f16")
    lg16 = m(imgs, widths=wd)
    mg, sc = _np_guard(lg16)
    assert np.array_equal(g["min_margin"], mg.astype(np.float32)) and np.array_equal(g["scale"], sc.astype(np.float32))
    thr = 2.0 * (LOGIT_RTOL * sc.astype(np.float64) + LOGIT_ATOL)
    assert np.array_equal(g["flags"].astype(bool), ~(mg.astype(np.float64) > thr))
    # logits in auto mode: flagged lines carry the f16x3 logits, the others the f16 ones, bit for bit
    m.set_precision("f16x3")
    lg3 = m(imgs, widths=wd)
    m.set_precision("auto")
    lga = m(imgs, widths=wd)
    assert np.array_equal(m.last_guard()["flags"], g["flags"])
    for b, f in enumerate(g["flags"]):
        assert np.array_equal(lga[:, b], lg3[:, b] if f else lg16[:, b]), (b, f)
    # an unflagged line's f16 argmax equals the f16x3 argmax on every column (what the criterion promises)
    for b, f in enumerate(g["flags"]):
        if not f:
            assert np.array_equal(lg16[:, b].argmax(axis=1), lg3[:, b].argmax(axis=1)), b


def test_guard_thresholds_and_mode_switching(pkg, synth, auto_random, state_dict):
    m = auto_random
    imgs = synth.make_line_images(3, 150, 77)
    t = _texts(pkg, synth, m, imgs)
    cd = pkg.ctc_codec(synth.characters())
    m.set_precision("auto")
    try:
        m.set_guard(0.0, 0.0)                         # only exact ties / NaNs are uncertain: nothing is re-run
        assert cd.labels_to_text(m.greedy(imgs)) == t["f16"] and m.last_guard()["flagged"] == 0
        m.set_guard(1e6, 0.0)                         # everything is uncertain: the f16x3 text
        assert cd.labels_to_text(m.greedy(imgs)) == t["f16x3"] and m.last_guard()["flagged"] == 3
    finally:
        m.set_guard(LOGIT_RTOL, LOGIT_ATOL)
    with pytest.raises(ValueError):
        m.set_guard(-1.0, 0.0)
    # after a call in another mode the figures are empty
    m.set_precision("f16")
    m.greedy(imgs)
    assert m.last_guard()["lines"] == 0
    # a context that built ONE weight set cannot move to a mode that needs the other
    m16 = pkg.hctr_model(synth.DEFAULT_VOCAB + 2).cuda(0)
    m16.load_state_dict(state_dict)
    with pytest.raises(RuntimeError):
        m16.set_precision("auto")
    with pytest.raises(RuntimeError):
        m16.set_precision("f16x3")
    m16.set_precision("f16")
    # sub-batching: the re-run of flagged lines in balanced passes gives the same labels as one pass
    m.set_precision("auto")
    m.set_guard(1e6, 0.0)
    try:
        big = synth.make_line_images(7, 150, 78)
        want = [x.tolist() for x in m.greedy(big)]
        os.environ["HCTR_MAX_COLS"] = "1000"           # (read at context creation)
        ms = pkg.hctr_model(synth.DEFAULT_VOCAB + 2, precision="auto").cuda(0)
        ms.load_state_dict(state_dict)
        ms.set_guard(1e6, 0.0)
        assert ms.lines_per_pass(7, 150, False) == 4 and ms.lines_per_pass(7, 150, True) == 2
        assert [x.tolist() for x in ms.greedy(big)] == want
        del ms
    finally:
        os.environ.pop("HCTR_MAX_COLS", None)
        m.set_guard(LOGIT_RTOL, LOGIT_ATOL)


def test_auto_mode_config2_trained_checkpoint_equals_the_real_reference(pkg, synth, auto_trained):
    """BASELINE configs[1] (64 x 1x128x2000), trained-like checkpoint: auto-mode text == the REAL reference's (fp32 CPU)
    for all 64 lines, with only a handful of lines run twice; the guard's margins agree with the reference's own margins
    (tests/golden/c2_trained_lines.npz) within twice the f16 tolerance."""
    with open(os.path.join(GOLDEN, "c2_trained_lines.json"), encoding="utf-8") as f:
        meta = json.load(f)
    g = np.load(os.path.join(GOLDEN, "c2_trained_lines.npz"))
    imgs = synth.make_font_lines(64, meta["width"], meta["seed"])
    m = auto_trained
    m.set_precision("auto")
    cd = pkg.ctc_codec(synth.characters())
    assert cd.labels_to_text(m.greedy(imgs)) == meta["greedy"]
    gd = m.last_guard()
    assert gd["lines"] == 64 and 0 <= gd["flagged"] <= 12, gd["flagged"]       # reference margins: 5-6 lines below the bound
    tol = LOGIT_RTOL * float(np.abs(g["max"]).max()) + LOGIT_ATOL
    assert np.abs(gd["min_margin"] - g["margin"].min(axis=1)).max() <= 2 * tol
    # the engine's margin is within 2 * tol of the reference's, so a line whose REFERENCE margins all exceed 4 * tol
    # cannot be flagged (the flag threshold is at most 2 * tol)
    ref_min = g["margin"].min(axis=1)
    assert not gd["flags"][ref_min > 4 * tol].any()


def test_auto_mode_config2_random_head_equals_f16x3(pkg, synth, auto_random):
    """The near-tie-rich random head at full size: every line has columns inside the tolerance, so every line is run
    again and the text is the f16x3 mode's on all 64 lines (59-60 of them the reference's: test_gpu_parity.py)."""
    with open(os.path.join(GOLDEN, "c2_lines.json"), encoding="utf-8") as f:
        meta = json.load(f)
    imgs = synth.make_line_images(64, 2000, meta["seed"])
    t = _texts(pkg, synth, auto_random, imgs)
    assert t["auto"] == t["f16x3"]
    assert t["guard"]["flagged"] == 64
    assert sum(a == b for a, b in zip(t["auto"], meta["greedy"])) >= 58


def test_auto_mode_ragged_config3_batch_equals_the_real_reference(pkg, synth, auto_trained):
    """The batch shape test.py -b N builds (test.py:170-186): four widths padded together with NormalizePAD's replicate
    pad. The pad region's logits are flat, f16 alone differs from the reference there (gpurun_out/r2m); the guard flags
    exactly such lines and the text EQUALS the reference's (tests/golden/c3_lines.json)."""
    with open(os.path.join(GOLDEN, "c3_lines.json"), encoding="utf-8") as f:
        gold = json.load(f)
    widths = gold["ragged"]["widths"]
    batch = np.zeros((len(widths), 128, max(widths)), np.uint8)
    for i, w in enumerate(widths):
        batch[i, :, :w] = synth.make_font_lines(1, w, gold["seed"], line_offset=gold["widths"].index(w) * 128)[0]
    wd = np.array(widths, np.int32)
    t = _texts(pkg, synth, auto_trained, batch, wd)
    assert t["auto"] == gold["ragged"]["greedy"]
    assert t["f16x3"] == gold["ragged"]["greedy"]
    flags = t["guard"]["flags"].astype(bool)
    for b in range(len(widths)):                     # wherever f16 alone is wrong the guard has caught the line
        if t["f16"][b] != gold["ragged"]["greedy"][b]:
            assert flags[b], b
    # bucket lines (equal widths, no pad): exact as well
    cd = pkg.ctc_codec(synth.characters())
    auto_trained.set_precision("auto")
    for bi, w in enumerate(gold["widths"]):
        imgs = synth.make_font_lines(2, w, gold["seed"], line_offset=bi * 128)
        assert cd.labels_to_text(auto_trained.greedy(imgs)) == gold["buckets"][str(w)], w


def test_auto_mode_config5_beam_strings_equal_the_real_reference(pkg, synth, auto_trained):
    """Beam front end in auto mode: flagged lines' top-k / blank / candidate lists are the f16x3 mode's, the others the
    f16 mode's; the strings equal the REAL reference's (tests/golden/c5_beam_lines.json) for all three variants."""
    with open(os.path.join(GOLDEN, "c5_beam_lines.json"), encoding="utf-8") as f:
        gold = json.load(f)
    m = auto_trained
    imgs = synth.make_font_lines(gold["lines"], gold["width"], gold["seed"])
    cd = pkg.ctc_codec(synth.characters()).attach(m)
    fes = {}
    for mode in ("f16", "f16x3", "auto"):
        m.set_precision(mode)
        fes[mode] = m.beam_frontend(imgs, k=10, want_candidates=True)
    flags = m.last_guard()["flags"].astype(bool)
    assert len(flags) == gold["lines"]
    B = gold["lines"]
    for b in range(B):
        src = fes["f16x3"] if flags[b] else fes["f16"]
        for key in ("topk_idx", "topk_logp", "blank_logp"):
            assert np.array_equal(fes["auto"][key][:, b], src[key][:, b]), (b, key)
        for t in range(0, gold["width"], 97):
            r = t * B + b
            lo, hi = fes["auto"]["cand_off"][r], fes["auto"]["cand_off"][r + 1]
            slo, shi = src["cand_off"][r], src["cand_off"][r + 1]
            assert np.array_equal(fes["auto"]["cand_idx"][lo:hi], src["cand_idx"][slo:shi])
            assert np.array_equal(fes["auto"]["cand_logp"][lo:hi], src["cand_logp"][slo:shi])
    m.set_precision("auto")
    assert cd.labels_to_text(m.greedy(imgs)) == gold["greedy"]
    for tag, skip, lm in (("full_toy", False, pkg.ToyBigramLM()), ("full_zero", False, pkg.ZeroLM()),
                          ("skip_toy", True, pkg.ToyBigramLM())):
        cd.use_beam_search, cd.skip_search, cd.use_tfm_pred, cd.use_tfm_score = True, skip, False, False
        cd.lm_panelty, cd.len_bonus, cd.beam_size, cd.search_depth, cd.ngram = 0.8, 4.8, 10, 10, lm
        fe = m.beam_frontend(imgs, k=10, want_candidates=skip)
        assert cd.decode_frontend(fe) == gold[tag], tag
    # a forced re-run of some lines (everything uncertain) gives the f16x3 front end for all lines
    m.set_guard(1e6, 0.0)
    try:
        fe = m.beam_frontend(imgs, k=10, want_candidates=True)
        for key in ("topk_idx", "topk_logp", "blank_logp", "cand_off"):
            assert np.array_equal(fe[key], fes["f16x3"][key]), key
        n = int(fe["cand_off"][-1])
        assert np.array_equal(fe["cand_idx"][:n], fes["f16x3"]["cand_idx"][:n])
    finally:
        m.set_guard(LOGIT_RTOL, LOGIT_ATOL)


def test_auto_mode_nan_and_ties_are_flagged(pkg, synth, state_dict):
    """a NaN logit or an exact tie can never be certified: such lines go to f16x3 (where np.argmax's order decides)"""
    sd = dict(state_dict)
    bias = state_dict["linear.bias"].copy()
    bias[5] = np.nan
    sd["linear.bias"] = bias
    C = synth.DEFAULT_VOCAB + 2
    m = pkg.hctr_model(C, precision="auto").cuda(0)
    m.load_state_dict(sd)
    imgs = synth.make_line_images(2, 80, 4)
    assert [lab.tolist() for lab in m.greedy(imgs)] == [[5], [5]]
    gd = m.last_guard()
    assert gd["flagged"] == 2 and np.isnan(gd["min_margin"]).all()
    # all logits equal: every column is an exact tie -> margin 0, flagged, argmax = class 0 (blank) -> empty text
    sd = dict(state_dict)
    sd["linear.weight"] = np.zeros_like(state_dict["linear.weight"])
    sd["linear.bias"] = np.full_like(state_dict["linear.bias"], 0.25)
    m2 = pkg.hctr_model(C, precision="auto").cuda(0)
    m2.load_state_dict(sd)
    assert [lab.tolist() for lab in m2.greedy(imgs)] == [[], []]
    gd = m2.last_guard()
    assert gd["flagged"] == 2 and (gd["min_margin"] == 0).all() and (gd["scale"] == 0.25).all()


def test_cli_precision_auto(tmp_path, pkg, synth, auto_random):
    """test.py --precision auto (the drop-in CLI with ragged files in batches of 2, i.e. padded batches): the printed
    strings are the f16x3 mode's for the same padded batches."""
    import ast
    import subprocess
    import sys
    from PIL import Image
    from conftest import ROOT
    folder = tmp_path / "lines"
    folder.mkdir()
    widths = [120, 64, 97, 97]
    imgs = [synth.make_line_images(1, w, 700 + i)[0] for i, w in enumerate(widths)]
    for i, im in enumerate(imgs):
        Image.fromarray(im).save(folder / ("%06d.png" % i))
    cd = pkg.ctc_codec(synth.characters())
    auto_random.set_precision("f16x3")
    want = []
    for i in range(0, 4, 2):
        w = max(widths[i:i + 2])
        batch = np.zeros((2, 128, w), np.uint8)
        for j in range(2):
            batch[j, :, :widths[i + j]] = imgs[i + j]
        want += cd.labels_to_text(auto_random.greedy(batch, widths=np.array(widths[i:i + 2], np.int32)))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "test.py"), "-m", "hctr", "-f", "synthetic", "-b", "2", "-i",
                        str(folder), "-dm", "greedy-search", "--precision", "auto"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = []
    for line in r.stdout.splitlines():
        if line.startswith("predicted results: "):
            got += ast.literal_eval(line[len("predicted results: "):])
    assert got == want
