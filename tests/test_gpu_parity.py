"""GPU parity tests (-m gpu): the HIP engine, called through the C ABI, against the oracle and the
committed golden fixtures (outputs of the real reference).

Two floating-point checks, both with the tolerance written here:
 (1) against the fp32 CPU reference (fixtures from the REAL reference, and the oracle). The engine
     stores activations/weights in fp16 and accumulates in fp32 (the reference's own GPU mode is
     TF32, the same 10-bit mantissa), so:
       logits  |err| <= LOGIT_RTOL * max|logit| + LOGIT_ATOL
       argmax  bit-exact on every column whose reference top-2 margin exceeds 2x that FIXED tolerance
               (a flip needs err(top1) + err(top2) > margin); the share of such columns is a property of
               the fixture and recorded per fixture (MIN_SAFE); >= MIN_AGREE of all columns agree
       text    character error rate vs the reference strings <= F16_CER_MAX over the 64 config-2 lines (fixed; measured
               2.4 %: their random-head logits have a near-tie in 1 column of 8), <= F16_LINE_CER_MAX for a single
               line; f16x3 mode and
               the trained-like checkpoint carry the exact-text assertions
 (2) against the oracle with the engine's rounding points inserted (oracle.hctr_ref.forward_f16):
     end to end |err| <= 0.008 * max|logit| (rounding-boundary flips still decorrelate two fp16
     pipelines), but PER LAYER, from the engine's own activations, every element within one fp16
     ulp and <= 3 % of elements differing at all; head logits within 2e-4 * max|logit|.
Integer / index work (collapse, top-k order, candidate lists, beam search) is bit-exact.
"""
import json
import os

import numpy as np
import pytest

import codec_cases
from conftest import GOLDEN
from oracle import ctc_ref, hctr_ref

pytestmark = pytest.mark.gpu

LOGIT_RTOL = 0.01
LOGIT_ATOL = 0.05
MIN_AGREE = 0.97        # measured 0.988 on the 128 000 columns of config 2 (gpurun r2a)
F16_CER_MAX = 0.04      # measured 0.024 there
# share of columns whose reference top-2 margin exceeds 2 * (LOGIT_RTOL * scale + LOGIT_ATOL): computed from the
# fixtures alone (tests/golden), recorded here so that the coverage of the argmax-exact assertion is explicit
MIN_SAFE = {"b1w32": 0.65, "b3w67u": 0.83, "b2w96": 0.82, "w488": 0.75, "w2000": 0.71, "b2w300u": 0.74,
            "b4w131u": 0.85, "c1": 0.83, "c2": 0.66}


def f16_tol(scale):
    return LOGIT_RTOL * float(scale) + LOGIT_ATOL


def check_f16_argmax(got_arg, ref_arg, margin, scale, min_safe):
    """argmax identical on every column the FIXED tolerance cannot flip; overall agreement floor."""
    safe = margin > 2 * f16_tol(scale)
    assert safe.mean() >= min_safe, safe.mean()
    assert np.array_equal(got_arg[safe], ref_arg[safe])
    assert (got_arg == ref_arg).mean() >= MIN_AGREE
    return safe


F16_LINE_CER_MAX = 0.07   # a single line (30-1000 characters) scatters around the 2.4 % aggregate: measured up to 5.2 %


def check_f16_text(mine, want):
    """fixed character-error-rate bound for ONE line (at least two edits are allowed on very short strings)"""
    bound = max(2, int(np.ceil(F16_LINE_CER_MAX * len(want))))
    assert ctc_ref.edit_distance(mine, want) <= bound, (ctc_ref.edit_distance(mine, want), bound, len(want))


@pytest.fixture(scope="module")
def engine(pkg, synth, state_dict):
    m = pkg.hctr_model(synth.DEFAULT_VOCAB + 2).cuda(0)
    m.load_state_dict(state_dict)
    m.eval()
    return m


@pytest.fixture(scope="module")
def codec(pkg, synth, engine):
    return pkg.ctc_codec(synth.characters()).attach(engine)


def _margin(ref):
    srt = np.sort(ref, axis=2)
    return srt[:, :, -1] - srt[:, :, -2]


@pytest.mark.parametrize("name,seed,widths", [("b1w32", 21, [32]), ("b3w67u", 22, [67, 50, 33]),
                                              ("b2w96", 23, [96, 96])])
def test_forward_matches_reference_fixture(engine, synth, name, seed, widths):
    """Engine logits vs the REAL reference's logits (tests/golden/model_small.npz)."""
    g = np.load(os.path.join(GOLDEN, "model_small.npz"))
    imgs = synth.make_line_images(len(widths), max(widths), seed)
    got = engine(imgs, widths=widths)
    sub = g["sub_classes"]
    ref_sub = g[name + "/logits_sub"]
    tol = LOGIT_RTOL * float(np.abs(ref_sub).max()) + LOGIT_ATOL
    err = float(np.abs(got[:, :, sub] - ref_sub).max())
    assert err <= tol, "max logit error %.4f > %.4f" % (err, tol)
    margin = g[name + "/top10_val"][:, :, 0] - g[name + "/top10_val"][:, :, 1]
    check_f16_argmax(got.argmax(axis=2), g[name + "/argmax"].astype(np.int64), margin, np.abs(ref_sub).max(),
                     MIN_SAFE[name])
    # activations along the trunk (debug taps) against the reference's
    # block1.0 = first block of stage 1: fused SE + the 1x1 downsample branch inside conv2's K loop (buffer p1.1);
    # block3.4 = last block of stage 3 (buffer p3.0)   [buffer rotation: engine.cpp run_forward]
    for tap, buf in (("stage0", "stage0"), ("stage2", "stage2"), ("stage4", "stage4"), ("block1.0", "p1.1"),
                     ("block3.4", "p3.0")):
        a = engine.debug_activation(buf, len(widths))[:, :8, :, :16]
        r = g[name + "/act/" + tap]
        assert np.abs(a - r).max() <= 0.02 * np.abs(r).max() + 0.02, tap


@pytest.mark.parametrize("seed,widths", [(22, [67, 50, 33]), (24, [130, 130])])
def test_forward_matches_f16_oracle(engine, synth, state_dict, seed, widths):
    """Engine vs the oracle with the engine's rounding points inserted.

    End to end the two fp16 pipelines still decorrelate (a one-ulp flip at a rounding boundary
    perturbs ~1000 downstream sums per layer), so the end-to-end bound is only ~2x tighter than
    against fp32. The TIGHT check is per layer: feed the engine's own tap into the oracle's next layer
    and require the engine's next tap to match to one fp16 ulp on all but a sliver of elements."""
    import torch
    import torch.nn.functional as F
    B = len(widths)
    imgs = synth.make_line_images(B, max(widths), seed)
    ref = hctr_ref.forward_f16(state_dict, synth.normalize_pad(imgs, widths)).numpy()
    got = engine(imgs, widths=widths)
    scale = float(np.abs(ref).max())
    assert float(np.abs(got - ref).max()) <= 0.008 * scale
    assert (got.argmax(axis=2) == ref.argmax(axis=2)).mean() >= 0.96    # statistical (decorrelated roundings)

    def check_layer(tap_in, tap_out, ck, bk, pool, what, half_weights=True, x_in=None):
        """engine[tap_out] vs oracle layer applied to engine[tap_in]. Bound per element:
        one fp16 ulp of the result + accumulation noise, which scales with the magnitude of the
        summed terms (sum |w||x|), not of the result (cancellation-heavy channels). Measured on
        gfx950: v_mfma_f32_16x16x32_f16 accumulation error grows ~linearly with K (about
        K * 2^-24 * sum|w||x| / K worst case, e.g. 7e-5 at K=576, sum|terms|~20), hence 2^-16."""
        xin = x_in if x_in is not None else torch.from_numpy(engine.debug_activation(tap_in, B))
        _, r = hctr_ref._conv_f16(state_dict, xin, ck, bk, True, 1, pool=pool, half_weights=half_weights)
        w, bias = hctr_ref._fold(state_dict, ck, bk, half_weights)
        mag = F.conv2d(xin.abs(), w.abs(), bias.abs(), padding=1)
        if pool:
            mag = F.max_pool2d(mag, (2, 1), (2, 1))
        r, mag = r.numpy(), mag.numpy()
        a = engine.debug_activation(tap_out, B)
        tol = np.abs(r) * 2.0 ** -10 + mag * 2.0 ** -16 + 1e-6
        diff = np.abs(a - r)
        assert (diff <= tol).all(), (what, float((diff / tol).max()))
        assert (diff > 0).mean() <= 0.03, what

    x = torch.from_numpy(synth.normalize_pad(imgs, widths))
    # the stem as two launches (HCTR_FUSE_STEM=0: conv0_1 stored, generic 64x256 MFMA conv with fused pool), layer by
    # layer; then the default fused kernel (conv0_1 computed into conv0_2's LDS halo) must reproduce it BIT FOR BIT
    default_engine = engine
    os.environ["HCTR_FUSE_STEM"] = "0"
    try:
        engine = type(default_engine)(synth.DEFAULT_VOCAB + 2).cuda(0)
    finally:
        del os.environ["HCTR_FUSE_STEM"]
    engine.load_state_dict(state_dict)
    assert np.array_equal(engine(imgs, widths=widths), got)
    check_layer(None, "conv0_1", "cnn.conv0_1", "cnn.bn0_1", False, "stem", half_weights=False, x_in=x)
    check_layer("conv0_1", "stage0", "cnn.conv0_2", "cnn.bn0_2", True, "conv0_2+pool")
    unfused_stage0 = engine.debug_activation("stage0", B)
    engine = default_engine
    engine(imgs, widths=widths)
    assert np.array_equal(engine.debug_activation("stage0", B), unfused_stage0)
    with pytest.raises(RuntimeError):
        engine.debug_activation("conv0_1", B)                      # fused: that tensor never exists
    # MFMA conv, 128x128 tile: block1.1.conv1 from the engine's own block1.0 output (buffer p1.1)
    check_layer("p1.1", "p1.2", "cnn.block1.1.conv1", "cnn.block1.1.bn1", False, "block1.1.conv1")
    # MFMA conv, 256x256 tile: block2.3.conv1 (p2.0) from block2.2's output (p2.2); block3.4.conv1
    # (p3.2) from block3.3's output (p3.1)  [buffer rotation: engine.cpp run_forward]
    check_layer("p2.2", "p2.0", "cnn.block2.3.conv1", "cnn.block2.3.bn1", False, "block2.3.conv1")
    check_layer("p3.1", "p3.2", "cnn.block3.4.conv1", "cnn.block3.4.bn1", False, "block3.4.conv1")
    # fused pool + head-input layout: conv4+pool from block4.0's output (p4.1)
    check_layer("p4.1", "stage4", "cnn.conv4", "cnn.bn4", True, "conv4+pool")
    # head GEMM from the engine's own pooled features
    f = torch.from_numpy(engine.debug_activation("stage4", B)).flatten(1, 2).permute(0, 2, 1)
    w = torch.from_numpy(state_dict["linear.weight"]).half().float()
    lg = F.linear(f, w, torch.from_numpy(state_dict["linear.bias"])).permute(1, 0, 2).numpy()
    assert float(np.abs(got - lg).max()) <= 2e-4 * scale          # fp32 summation order only
    assert (got.argmax(axis=2) == lg.argmax(axis=2)).mean() >= 0.999


def test_input_paths_agree(engine, synth):
    """uint8 + widths (device NormalizePAD) == float32 pre-normalised input (utils/dataset.py:83-93)."""
    widths = [80, 41, 80, 7]
    imgs = synth.make_line_images(4, 80, 77)
    a = engine(imgs, widths=widths)
    b = engine(synth.normalize_pad(imgs, widths))
    assert np.array_equal(a, b)
    import torch
    c = engine(torch.from_numpy(synth.normalize_pad(imgs, widths)).cuda())
    assert c.is_cuda and np.array_equal(c.cpu().numpy(), a)


def test_fused_greedy_equals_decode_of_logits(engine, codec, synth):
    """hctr_greedy (fused) == argmax+collapse of the engine's own logits == oracle codec on them:
    integer work, bit-exact."""
    imgs = synth.make_line_images(3, 200, 5)
    logits = engine(imgs)
    fused = codec.labels_to_text(engine.greedy(imgs))
    assert fused == codec.decode(logits)
    assert fused == ctc_ref.CtcCodecRef(synth.characters()).decode(logits)


@pytest.mark.parametrize("name,seed,w", [("w488", 31, 488), ("w2000", 32, 2000)])
def test_long_line_against_reference_fixture(engine, codec, synth, name, seed, w):
    """Config-2 width (2000) and a bundled-image width (488): argmax on safe columns and decoded text
    vs the real reference (tests/golden/model_lines.npz, model_strings.json)."""
    g = np.load(os.path.join(GOLDEN, "model_lines.npz"))
    with open(os.path.join(GOLDEN, "model_strings.json")) as f:
        strings = json.load(f)
    imgs = synth.make_line_images(1, w, seed)
    got = engine(imgs)
    err = float(np.abs(got.max(axis=2) - g[name + "/max"]).max())
    tol = LOGIT_RTOL * float(np.abs(g[name + "/max"]).max()) + LOGIT_ATOL
    assert err <= tol
    margin = g[name + "/top10_val"][:, :, 0] - g[name + "/top10_val"][:, :, 1]
    check_f16_argmax(got.argmax(axis=2), g[name + "/argmax"].astype(np.int64), margin, np.abs(g[name + "/max"]).max(),
                     MIN_SAFE[name])
    text = codec.labels_to_text(engine.greedy(imgs))[0]
    check_f16_text(text, strings[name]["greedy"][0])


def test_config1_bundled_images(engine, codec):
    """BASELINE config 1: the five bundled sample lines (W = 3514, 908, 2375, 1913, 488)."""
    g = np.load(os.path.join(GOLDEN, "images_c1.npz"))
    with open(os.path.join(GOLDEN, "model_strings.json")) as f:
        strings = json.load(f)
    for key in sorted({k.split("/")[0] for k in g.files}):
        img = g[key + "/image"]
        got = engine(img[None])
        err = float(np.abs(got.max(axis=2) - g[key + "/max"]).max())
        assert err <= LOGIT_RTOL * float(np.abs(g[key + "/max"]).max()) + LOGIT_ATOL, key
        check_f16_argmax(got.argmax(axis=2), g[key + "/argmax"].astype(np.int64), g[key + "/top2_margin"],
                         np.abs(g[key + "/max"]).max(), MIN_SAFE["c1"])
        text = codec.labels_to_text(engine.greedy(img[None]))[0]
        check_f16_text(text, strings["c1_" + key]["greedy"][0])


def test_codec_cases_on_device(pkg):
    """ctc_codec.decode on caller logits: greedy + both beam variants, vs the REAL reference codec's
    outputs (tests/golden/codec_cases.json). Bit-exact (strings)."""
    with open(os.path.join(GOLDEN, "codec_cases.json")) as f:
        gold = json.load(f)
    for name, seed, w, b, c, style in codec_cases.CODEC_CASES:
        logits = codec_cases.gen_logits(seed, w, b, c, style)
        chars = codec_cases.vocab(c)
        cd = pkg.ctc_codec(chars)
        assert cd.decode(logits) == gold[name]["greedy"], name
        for tag, skip, lm, lp, lb, bs, depth in codec_cases.BEAM_SETTINGS:
            cd = pkg.ctc_codec(chars)
            cd.use_beam_search, cd.skip_search, cd.use_tfm_pred = True, skip, False
            cd.lm_panelty, cd.len_bonus, cd.beam_size, cd.search_depth = lp, lb, bs, depth
            cd.ngram = pkg.ZeroLM() if lm == "zero" else pkg.ToyBigramLM()
            try:
                got = cd.decode(logits)
            except IndexError:
                got = "IndexError"
            assert got == gold[name][tag], (name, tag)
        # the callback path (a user LM object) must give the same strings as the built-in LM
        cd = pkg.ctc_codec(chars)
        cd.use_beam_search, cd.use_tfm_pred = True, False
        cd.lm_panelty, cd.len_bonus = 0.8, 4.8
        cd.ngram = ctc_ref.ToyBigramLM()
        assert cd.decode(logits) == gold[name]["full_toy"], name


def test_beam_on_model_logits(engine, pkg, synth):
    """Beam search end to end (fused front end) vs the oracle codec run on the engine's own logits
    (isolates decode parity from fp16 logit differences), and vs the reference fixture strings."""
    imgs = synth.make_line_images(2, 160, 41)
    logits = engine(imgs)
    for tag, skip, lm, lp, lb, bs, depth in codec_cases.BEAM_SETTINGS[:4]:
        cd = pkg.ctc_codec(synth.characters()).attach(engine)
        cd.use_beam_search, cd.skip_search, cd.use_tfm_pred = True, skip, False
        cd.lm_panelty, cd.len_bonus, cd.beam_size, cd.search_depth = lp, lb, bs, depth
        cd.ngram = pkg.ZeroLM() if lm == "zero" else pkg.ToyBigramLM()
        fe = engine.beam_frontend(imgs, k=depth, want_candidates=skip)
        fused = cd.decode_frontend(fe)
        assert fused == cd.decode(logits), tag
        oc = ctc_ref.CtcCodecRef(synth.characters())
        oc.use_beam_search, oc.skip_search, oc.use_tfm_pred = True, skip, False
        oc.lm_panelty, oc.len_bonus, oc.beam_size, oc.search_depth = lp, lb, bs, depth
        oc.ngram = ctc_ref.ZeroLM() if lm == "zero" else ctc_ref.ToyBigramLM()
        assert fused == oc.decode(logits), tag


def test_error_behaviour(pkg, synth, state_dict):
    m = pkg.hctr_model(synth.DEFAULT_VOCAB + 2)
    with pytest.raises(RuntimeError):
        m(np.zeros((1, 1, 128, 32), np.float32))          # never moved to a GPU: no CPU fallback
    m.cuda(0)
    bad = dict(state_dict)
    bad.pop("cnn.block2.1.se.fc.0.weight")
    with pytest.raises(KeyError):
        m.load_state_dict(bad)
    m2 = pkg.hctr_model(synth.DEFAULT_VOCAB + 2).cuda(0)
    bad = dict(state_dict)
    bad["extra.weight"] = np.zeros((1,), np.float32)
    with pytest.raises(KeyError):
        m2.load_state_dict(bad)
    m3 = pkg.hctr_model(100).cuda(0)
    with pytest.raises(RuntimeError):
        m3.load_state_dict(state_dict)                     # linear.weight shape mismatch


def test_bucketed_mixed_widths_and_subbatches(pkg, engine, codec, synth, state_dict):
    """Config-3 style: mixed widths through the bucketing helper; every bucket equals the oracle run on
    that bucket with NormalizePAD semantics, and a batch split into internal passes (HCTR_MAX_COLS
    forced small through a second context) gives bit-identical labels to the single-pass run."""
    from importlib import import_module
    import os
    bk = import_module(pkg.__name__ + ".bucketing")
    rng_w = [96, 96, 40, 41, 150, 147, 96, 33]
    images = [synth.make_line_images(1, w, 300 + i)[0] for i, w in enumerate(rng_w)]
    texts = bk.recognize(engine, codec, images, max_lines=3, max_pad_fraction=0.1)
    assert all(isinstance(t, str) for t in texts)
    oc = ctc_ref.CtcCodecRef(synth.characters())
    for idx in bk.plan_batches(rng_w, 3, 0.1):
        batch, ws = bk.pad_batch(images, idx)
        logits = engine(batch, widths=ws)
        ref = hctr_ref.forward(state_dict, synth.normalize_pad(batch, ws)).numpy()
        assert np.abs(logits - ref).max() <= LOGIT_RTOL * np.abs(ref).max() + LOGIT_ATOL
        assert [texts[i] for i in idx] == oc.decode(logits)          # bucketed result == per-batch result
    # internal sub-batching: same labels whether the batch runs in one pass or several
    imgs = synth.make_line_images(6, 64, 9)
    one = engine.greedy(imgs)
    os.environ["HCTR_MAX_COLS"] = "130"                              # 2 lines per pass at W=64
    try:
        small = pkg.hctr_model(synth.DEFAULT_VOCAB + 2).cuda(0)
        small.load_state_dict(state_dict)
    finally:
        del os.environ["HCTR_MAX_COLS"]
    many = small.greedy(imgs)
    assert all(np.array_equal(a, b) for a, b in zip(one, many))
    assert np.array_equal(small(imgs), engine(imgs))


def test_cli_reference_flags(tmp_path, engine, codec, synth):
    """test.py drop-in CLI (reference flags test.py:24-106): folder inference with greedy and beam
    decode, and -bm benchmark mode computing CER over <input>/test_img_id_gt.txt."""
    import ast
    import subprocess
    import sys
    from PIL import Image
    from conftest import ROOT
    data = tmp_path / "data"
    (data / "test").mkdir(parents=True)
    widths = [90, 64, 90, 77]
    imgs = [synth.make_line_images(1, w, 500 + i)[0] for i, w in enumerate(widths)]
    want = []
    for i, im in enumerate(imgs):
        Image.fromarray(im).save(data / "test" / ("%06d.png" % i))
    for i in range(0, 4, 2):                                  # the CLI batches files in sorted order, -b 2
        w = max(widths[i:i + 2])
        batch = np.zeros((2, 128, w), np.uint8)
        for j in range(2):
            batch[j, :, :widths[i + j]] = imgs[i + j]
        want += codec.labels_to_text(engine.greedy(batch, widths=np.array(widths[i:i + 2], np.int32)))
    with open(data / "test_img_id_gt.txt", "w", encoding="utf-8") as f:
        for i, t in enumerate(want):
            f.write("%06d.png,%s\n" % (i, t if i != 3 else t[:-1] + "?"))     # one deliberate error
    base = [sys.executable, os.path.join(ROOT, "test.py"), "-m", "hctr", "-f", "synthetic", "-b", "2"]
    r = subprocess.run(base + ["-i", str(data / "test"), "-dm", "greedy-search"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = []
    for line in r.stdout.splitlines():
        if line.startswith("predicted results: "):
            got += ast.literal_eval(line[len("predicted results: "):])
    assert got == want
    r = subprocess.run(base + ["-i", str(data), "-dm", "greedy-search", "-bm"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    cer = float([l for l in r.stdout.splitlines() if l.startswith("Total Test CER:")][0].split()[3])
    nchar = sum(len(t) for t in want)
    assert abs(cer - 1.0 / nchar) < 1e-9
    r = subprocess.run(base + ["-i", str(data / "test"), "-dm", "beam-search", "-kp", "toy", "-ss", "-bs", "5"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("predicted results: ") == 2
    # half-height files: INTER_AREA enlarging x2 replicates pixels (device resize inside the CLI)
    small = tmp_path / "small"
    small.mkdir()
    halves = [im[::2, ::2] for im in imgs[:2]]
    for i, im in enumerate(halves):
        Image.fromarray(im).save(small / ("%06d.png" % i))
    up = [np.repeat(np.repeat(h, 2, axis=0), 2, axis=1) for h in halves]
    w = max(u.shape[1] for u in up)
    batch = np.zeros((2, 128, w), np.uint8)
    for j, u in enumerate(up):
        batch[j, :, :u.shape[1]] = u
    want_up = codec.labels_to_text(engine.greedy(batch, widths=np.array([u.shape[1] for u in up], np.int32)))
    r = subprocess.run(base + ["-i", str(small), "-dm", "greedy-search"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = [ln for ln in r.stdout.splitlines() if ln.startswith("predicted results: ")]
    assert ast.literal_eval(got[0][len("predicted results: "):]) == want_up
    arpa = codec_cases.write_toy_arpa(str(tmp_path / "toy.arpa"))                  # -kp model.arpa: native LM
    r = subprocess.run(base + ["-i", str(data / "test"), "-dm", "beam-search", "-kp", arpa], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("predicted results: ") == 2


@pytest.mark.parametrize("B,W", [(1, 1), (2, 5), (1, 16), (3, 17)])
def test_tiny_and_odd_widths(engine, codec, synth, state_dict, B, W):
    """Edge shapes: widths below / just above the 16-column tile, single column."""
    imgs = synth.make_line_images(B, max(W, 8), 900 + W)[:, :, :W]
    imgs = np.ascontiguousarray(imgs)
    got = engine(imgs)
    ref = hctr_ref.forward(state_dict, synth.normalize_pad(imgs)).numpy()
    assert got.shape == ref.shape == (W, B, synth.DEFAULT_VOCAB + 2)
    assert np.abs(got - ref).max() <= LOGIT_RTOL * np.abs(ref).max() + LOGIT_ATOL
    fused = codec.labels_to_text(engine.greedy(imgs))
    assert fused == codec.decode(got)


def test_empty_batch(engine, synth):
    out = engine(np.zeros((0, 128, 40), np.uint8))
    assert out.shape == (40, 0, synth.DEFAULT_VOCAB + 2)
    assert engine.greedy(np.zeros((0, 128, 40), np.uint8)) == []


def test_other_vocabulary_size_and_checkpoint_file(tmp_path, pkg, synth):
    """A 100-character vocabulary (C=102, head padded to 256) loaded from a real ``.pth.tar`` checkpoint
    file in the reference's format (main.py:349-356), through the model API and through test.py."""
    import subprocess
    import sys
    import torch
    from PIL import Image
    from conftest import ROOT
    C = 102
    sd = synth.make_state_dict(C, seed=0)        # seed 0: the BN calibration data belongs to this trunk
    ckpt = tmp_path / "hctr_checkpoint.pth.tar"
    torch.save({"epoch": 1, "state_dict": synth.to_torch(sd), "best_acc": 0.0, "optimizer": {}}, str(ckpt))
    m = pkg.hctr_model(C).cuda(0)
    m.load_state_dict(torch.load(str(ckpt), map_location="cpu", weights_only=True)["state_dict"])
    imgs = synth.make_line_images(2, 70, 77)
    got = m(imgs)
    ref = hctr_ref.forward(sd, synth.normalize_pad(imgs)).numpy()
    assert got.shape == (70, 2, C)
    assert np.abs(got - ref).max() <= LOGIT_RTOL * np.abs(ref).max() + LOGIT_ATOL
    chars = synth.characters(C - 2)
    cd = pkg.ctc_codec(chars).attach(m)
    want = cd.labels_to_text(m.greedy(imgs[:1]))
    data = tmp_path / "set" / "test"
    data.mkdir(parents=True)
    Image.fromarray(imgs[0]).save(data / "a.png")
    with open(tmp_path / "set" / "chars_list.txt", "w", encoding="utf-8") as f:   # test.py:315-326 discovery
        f.write(chars + "\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "test.py"), "-m", "hctr", "-f", str(ckpt), "-i", str(data),
                        "-dm", "greedy-search"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Model output classes: 102" in r.stdout
    assert ("predicted results: %r" % want) in r.stdout


def test_pipelined_beam_equals_sequential(pkg, engine, synth):
    """pipeline.recognize_beam (front end of chunk i+1 overlapped with the host search of chunk i)
    returns exactly what the two stages give back to back on the same chunks."""
    from importlib import import_module
    pipe = import_module(pkg.__name__ + ".pipeline")
    imgs = synth.make_line_images(5, 120, 61)
    cd = pkg.ctc_codec(synth.characters()).attach(engine)
    cd.use_beam_search, cd.use_tfm_pred, cd.ngram = True, False, pkg.ToyBigramLM()
    cd.lm_panelty, cd.len_bonus = 0.8, 4.8
    want = []
    for lo in range(0, 5, 2):
        want += cd.decode_frontend(engine.beam_frontend(imgs[lo:lo + 2], k=10))
    assert pipe.recognize_beam(engine, cd, imgs, chunk=2) == want


@pytest.fixture(scope="module")
def engine_x3(pkg, synth, state_dict):
    m = pkg.hctr_model(synth.DEFAULT_VOCAB + 2, precision="f16x3").cuda(0)
    m.load_state_dict(state_dict)
    return m


X3_RTOL = 2e-4          # f16x3 logits vs the fp32 CPU reference: |err| <= X3_RTOL * max|logit|


@pytest.mark.parametrize("name,seed,widths", [("b3w67u", 22, [67, 50, 33]), ("b2w96", 23, [96, 96])])
def test_f16x3_mode_matches_reference_closely(engine_x3, pkg, synth, name, seed, widths):
    """Split-precision mode (hi+lo fp16 pairs, three MFMA products per term) against the REAL reference's
    fixtures: logits to 2e-4 of the logit scale, argmax identical wherever the reference's own top-2 margin
    exceeds twice that, decoded text exact when the line has no such near-tie."""
    g = np.load(os.path.join(GOLDEN, "model_small.npz"))
    with open(os.path.join(GOLDEN, "model_strings.json")) as f:
        strings = json.load(f)
    imgs = synth.make_line_images(len(widths), max(widths), seed)
    got = engine_x3(imgs, widths=widths)
    sub = g["sub_classes"]
    ref_sub = g[name + "/logits_sub"]
    scale = float(np.abs(ref_sub).max())
    err = float(np.abs(got[:, :, sub] - ref_sub).max())
    assert err <= X3_RTOL * scale, (err, scale)
    assert float(np.abs(got.max(axis=2) - g[name + "/max"]).max()) <= X3_RTOL * scale
    margin = g[name + "/top10_val"][:, :, 0] - g[name + "/top10_val"][:, :, 1]
    safe = margin > 2 * X3_RTOL * scale
    ref_arg = g[name + "/argmax"].astype(np.int64)
    assert safe.mean() >= 0.98
    assert np.array_equal(got.argmax(axis=2)[safe], ref_arg[safe])
    cd = pkg.ctc_codec(synth.characters()).attach(engine_x3)
    text = cd.labels_to_text(engine_x3.greedy(imgs, widths=widths))
    for b in range(len(widths)):
        if safe[:, b].all():
            assert text[b] == strings[name]["greedy"][b]
    for tap in ("stage0", "stage2", "stage4"):
        a = engine_x3.debug_activation(tap, len(widths))[:, :8, :, :16]
        r = g[name + "/act/" + tap]
        assert np.abs(a - r).max() <= X3_RTOL * max(1.0, np.abs(r).max()), tap


def test_f16x3_long_line_text(engine_x3, pkg, synth):
    """W = 2000 (config-2 width): in f16x3 mode the greedy text EQUALS the fp32 CPU reference's
    (tests/golden/model_strings.json), vs ~2.4 % CER in the default f16 mode."""
    g = np.load(os.path.join(GOLDEN, "model_lines.npz"))
    with open(os.path.join(GOLDEN, "model_strings.json")) as f:
        strings = json.load(f)
    imgs = synth.make_line_images(1, 2000, 32)
    got = engine_x3(imgs)
    scale = float(np.abs(g["w2000/max"]).max())
    assert float(np.abs(got.max(axis=2) - g["w2000/max"]).max()) <= X3_RTOL * scale
    agree = (got.argmax(axis=2) == g["w2000/argmax"].astype(np.int64)).mean()
    assert agree >= 0.999
    cd = pkg.ctc_codec(synth.characters()).attach(engine_x3)
    text = cd.labels_to_text(engine_x3.greedy(imgs))[0]
    assert text == strings["w2000"]["greedy"][0]          # exact (this line has no column with a margin below 2e-4)


def test_model_moves_and_attached_codec(pkg, synth, state_dict):
    """nn.Module-like device moves keep the weights; an attached codec follows the model's context."""
    m = pkg.hctr_model(synth.DEFAULT_VOCAB + 2).cuda(0)
    m.load_state_dict(state_dict)
    cd = pkg.ctc_codec(synth.characters()).attach(m)
    imgs = synth.make_line_images(1, 48, 71)
    a = m(imgs)
    txt = cd.decode(a)
    m.cpu()
    with pytest.raises(RuntimeError):
        m(imgs)
    with pytest.raises(RuntimeError):
        cd.decode(a)                       # attached model is off the GPU: no dangling context
    m.cuda(0)                              # weights come back without another load_state_dict
    assert np.array_equal(m(imgs), a)
    assert cd.decode(a) == txt


def test_full_size_config2_properties(engine, codec, synth):
    """BASELINE config 2 at full size (B=64 x 1x128x2000, one 33 GB pass): properties that need no oracle.
    Lines are independent (eval-mode BN, per-image SE), so a line's labels do not depend on its batch;
    the forward is deterministic (no float atomics); fused greedy == argmax+collapse of the logits."""
    B, W = 64, 2000
    imgs = synth.make_line_images(B, W, 2)
    labels = engine.greedy(imgs)
    again = engine.greedy(imgs)
    assert all(np.array_equal(a, b) for a, b in zip(labels, again))               # determinism
    assert all(0 < len(l) <= W for l in labels)
    unk = synth.DEFAULT_VOCAB + 1
    for lab in labels[:8]:
        assert lab.min() >= 1 and lab.max() < unk                                  # never blank / <unknown>
    for i in (0, 17, 63):                                                          # batch invariance
        alone = engine.greedy(imgs[i:i + 1])[0]
        assert np.array_equal(alone, labels[i]), i
    pair = engine.greedy(imgs[[5, 40]])
    assert np.array_equal(pair[0], labels[5]) and np.array_equal(pair[1], labels[40])
    sub = imgs[:4]
    logits = engine(sub)                                                           # [2000, 4, 7358]
    assert codec.decode(logits) == codec.labels_to_text(labels[:4])               # fused == decode(logits)
    assert np.isfinite(logits).all()


@pytest.mark.parametrize("env", [{"HCTR_HALO": "0"}, {"HCTR_HALO": "1"}, {"HCTR_HALO": "0", "HCTR_PIPE": "1"},
                                 {"HCTR_FUSE_SE": "0"}, {"HCTR_HALO": "0", "HCTR_BIG_TILES": "0"},
                                 {"HCTR_PERSIST": "1"}, {"HCTR_PERSIST": "2"}, {"HCTR_FUSE_ARGMAX": "0"}, {"HCTR_FUSE_DS": "0"},
                                 {"HCTR_FUSE_STEM": "0"}, {"HCTR_WS_ALIAS": "1"}, {"HCTR_HALFHALO": "1"}, {"HCTR_RESPRE": "1"}], ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_alternative_kernel_paths(env):
    """The A/B kernels (generic 64x256/128x128/256x256 tiles, 8-wave halo, interleaved pipe, unfused SE,
    persistent tiles) stay correct: same fixture and tolerances as the default path. Kernel selection is
    read once per process, hence one child process per variant (run one after the other)."""
    import subprocess
    import sys
    from conftest import ROOT
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_altpath_check.py")], env=e,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_device_resize_bit_exact(engine, pkg, codec):
    """csrc/preprocess.hip (through hctr_resize_lines) vs oracle/resize_ref.py: every byte equal, for all three
    OpenCV INTER_AREA regimes (area, integer decimation, enlarging), gray / BGR / RGB sources, ragged batches,
    one-pixel sources, the AlignCollate crop; the device-resident batch decodes like the host one."""
    from oracle import resize_ref
    pp = pkg.preprocess
    rng = np.random.default_rng(77)
    shapes = [(48, 131), (53, 37), (77, 115), (127, 90), (128, 33), (129, 300), (200, 777), (256, 154), (384, 60),
              (300, 41), (1, 9), (2, 2), (500, 31), (64, 1), (131, 257), (640, 1000)]
    for rule in ("test", "dataset"):
        imgs = [rng.integers(0, 256, hw, dtype=np.uint8) for hw in shapes]
        got, widths = pp.resize_lines(engine, imgs, rule=rule)
        assert got.dtype == np.uint8 and got.shape[:2] == (len(imgs), 128)
        for i, im in enumerate(imgs):
            want = resize_ref.read_resize(im, 128, rule)
            assert widths[i] == want.shape[1]
            assert np.array_equal(got[i, :, :widths[i]], want), (rule, shapes[i])
            assert not got[i, :, widths[i]:].any()
    # colour sources, both channel orders
    col = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in ((60, 90), (200, 333), (128, 64))]
    for order in ("rgb", "bgr"):
        got, widths = pp.resize_lines(engine, col, order=order)
        for i, im in enumerate(col):
            assert np.array_equal(got[i, :, :widths[i]], resize_ref.read_resize(im, 128, "test", order))
    # AlignCollate crop (utils/dataset.py:118-145)
    wide = [rng.integers(0, 256, (64, 1000), dtype=np.uint8), rng.integers(0, 256, (64, 100), dtype=np.uint8)]
    got, widths = pp.resize_lines(engine, wide, rule="dataset", max_width=1600)
    assert got.shape == (2, 128, 1600) and widths.tolist() == [1600, 200]
    assert np.array_equal(got[0], resize_ref.read_resize(wide[0], 128, "dataset")[:, :1600])
    # device-resident output feeds the engine directly
    lines = [rng.integers(0, 256, (50 + 7 * i, 260 + 40 * i), dtype=np.uint8) for i in range(3)]
    host, widths = pp.resize_lines(engine, lines)
    dev, widths_d = pp.resize_lines(engine, lines, device_out=True)
    assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), host) and np.array_equal(widths, widths_d)
    a = engine.greedy(host, widths=widths)
    b = engine.greedy(dev, widths=widths)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # errors: cv2.resize raises on an empty destination, the shim raises ValueError
    with pytest.raises(ValueError):
        pp.resize_lines(engine, [np.zeros((500, 3), np.uint8)])           # int(128 * 3 / 500) == 0
    with pytest.raises(ValueError):
        pp.resize_lines(engine, [np.zeros((4, 4), np.float32)])
    assert pp.resize_lines(engine, [])[0].shape == (0, 128, 0)


def test_rccl_gather_path_runs():
    """bench.py's N>1 result gather through RCCL (backend "nccl"), with the one rank this box has: process
    group init bound to the device, dist.gather of device tensors, barrier. (World size 2 logic: gloo CPU test.)"""
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_nccl_gather_check.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok nccl gather" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_stamped_diagnostic_instance_keeps_results(engine, pkg, synth):
    """hctr_debug_stamps arms a separate kernel instance for one layer; the forward's results must not change
    and the stamps must be ordered (entry <= prologue <= landed <= K loop <= epilogue <= drained)."""
    import ctypes
    lib = pkg._lib.load()
    imgs = synth.make_line_images(2, 96, 23)
    want = engine(imgs)
    cap = 4096
    assert int(lib.hctr_debug_stamps(engine._ctx, b"block3.1.conv2+se", None, cap)) == 0
    try:
        got = engine(imgs)
        out = np.zeros((cap, 16), np.uint64)
        n = int(lib.hctr_debug_stamps(engine._ctx, None, out.ctypes.data_as(ctypes.c_void_p), cap))
    finally:
        lib.hctr_debug_stamps(engine._ctx, b"", None, 0)          # disarm
    assert np.array_equal(got, want)
    assert n == 2 * 1 * 6 * 4                                      # images x tile rows x tile columns x cout tiles
    t = out[:n, :6].astype(np.int64)
    assert (np.diff(t, axis=1) >= 0).all() and (t[:, 5] - t[:, 0]).max() < 10 ** 7


@pytest.mark.parametrize("name,seed,widths", [("b2w300u", 51, [300, 211]), ("b4w131u", 52, [131, 100, 64, 17])])
def test_forward_matches_reference_extra_fixtures(engine, codec, synth, name, seed, widths):
    """Engine vs the REAL reference on strongly unequal widths (tests/golden/model_extra.npz): logits, argmax on
    safe columns, trunk activations incl. the fused-downsample block, greedy text within the ambiguous-column bound."""
    g = np.load(os.path.join(GOLDEN, "model_extra.npz"))
    with open(os.path.join(GOLDEN, "model_extra_strings.json"), encoding="utf-8") as f:
        strings = json.load(f)[name]
    imgs = synth.make_line_images(len(widths), max(widths), seed)
    got = engine(imgs, widths=widths)
    ref_sub = g[name + "/logits_sub"]
    tol = LOGIT_RTOL * float(np.abs(ref_sub).max()) + LOGIT_ATOL
    err = float(np.abs(got[:, :, g["sub_classes"]] - ref_sub).max())
    assert err <= tol, "max logit error %.4f > %.4f" % (err, tol)
    margin = g[name + "/top10_val"][:, :, 0] - g[name + "/top10_val"][:, :, 1]
    check_f16_argmax(got.argmax(axis=2), g[name + "/argmax"].astype(np.int64), margin, np.abs(ref_sub).max(),
                     MIN_SAFE[name])
    for tap, buf in (("stage1", "stage1"), ("stage3", "stage3"), ("block1.0", "p1.1"), ("block3.4", "p3.0")):
        a = engine.debug_activation(buf, len(widths))[:, :8, :, :16]
        r = g[name + "/act/" + tap]
        assert np.abs(a - r).max() <= 0.02 * np.abs(r).max() + 0.02, tap
    text = codec.labels_to_text(engine.greedy(imgs, widths=widths))
    for mine, want in zip(text, strings["greedy"]):
        check_f16_text(mine, want)


def test_plain_c_caller_matches_python(tmp_path, engine, synth):
    """examples/greedy_demo.c - C99, no Python, no torch - drives the C ABI directly (weights from
    tools/export_weights.py) and must print the labels the Python shim returns."""
    import shutil
    import subprocess
    import sys
    from conftest import ROOT
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    libdir = os.path.join(ROOT, "handwritten-chinese-ocr-samples_amd")
    exe = str(tmp_path / "greedy_demo")
    r = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "greedy_demo.c"),
                        "-L", libdir, "-lhctr_hip", "-Wl,-rpath," + libdir, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    wfile = str(tmp_path / "weights.bin")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "export_weights.py"), "synthetic", wfile],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    W = 333
    img = synth.make_line_images(1, W, 91)
    img[0].tofile(str(tmp_path / "line.u8"))
    r = subprocess.run([exe, wfile, str(synth.DEFAULT_VOCAB + 2), str(tmp_path / "line.u8"), str(W)], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    want = engine.greedy(img)[0]
    line = [ln for ln in r.stdout.splitlines() if "labels:" in ln][0]
    got = [int(t) for t in line.split("labels:")[1].split()]
    assert got == want.tolist() and int(line.split()[0]) == len(want)


def test_two_contexts_on_two_threads(pkg, engine, synth, state_dict):
    """Threading contract of the C ABI: a context is not re-entrant, but independent contexts may
    run concurrently from different host threads (ctypes releases the GIL) - results equal the serial ones."""
    import threading
    other = pkg.hctr_model(synth.DEFAULT_VOCAB + 2).cuda(0)
    other.load_state_dict(state_dict)
    jobs = [(engine, synth.make_line_images(3, 300, 71)), (other, synth.make_line_images(2, 411, 72))]
    serial = [m.greedy(x) for m, x in jobs]
    out = [None, None]
    errs = []

    def work(i):
        try:
            m, x = jobs[i]
            for _ in range(4):
                out[i] = m.greedy(x)
        except BaseException as exc:                          # noqa: BLE001
            errs.append(exc)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for got, want in zip(out, serial):
        assert all(np.array_equal(a, b) for a, b in zip(got, want))


def test_kernel_families_agree_over_random_shapes(tmp_path):
    """30 random (lines, width, per-line widths) cases - widths below one tile, at tile edges, ragged batches -
    through the default path (halo kernels, fused SE / downsample / argmax) and through the independent generic
    kernels with every fusion off: logits within fp16-pipeline noise (1.2 % of scale), argmax agreement >= 93 %,
    and in each run the fused greedy result equals the decode of that run's own logits (tools/gpu_shape_sweep.py)."""
    import subprocess
    import sys
    from conftest import ROOT
    tool = os.path.join(ROOT, "tools", "gpu_shape_sweep.py")
    a, b = str(tmp_path / "a.npz"), str(tmp_path / "b.npz")
    base = {k: v for k, v in os.environ.items() if not k.startswith("HCTR_")}
    for path, extra in ((a, {}), (b, {"HCTR_HALO": "0", "HCTR_FUSE_SE": "0", "HCTR_FUSE_DS": "0", "HCTR_FUSE_ARGMAX": "0", "HCTR_FUSE_STEM": "0"})):
        env = dict(base)
        env.update(extra)
        r = subprocess.run([sys.executable, tool, "dump", path], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    r = subprocess.run([sys.executable, tool, "compare", a, b], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout[-1500:] + r.stderr[-1500:]


def test_two_ranks_equal_one_rank():
    """the same 6-line unequal-width batch decoded by 1 rank and by 2 ranks (contiguous shards,
    global pad width fixed before sharding, ONE gather) gives identical label arrays in identical order. Two
    processes share this box's GPU; the gather runs over gloo (RCCL needs one GPU per rank)."""
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29517",
                        os.path.join(ROOT, "tests", "dist_identity_worker.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "IDENTITY_OK world=2 lines=6" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_bench_strong_scaling_rehearsal():
    """bench.py --gpus 2 from a PLAIN command line (no launcher, no rank environment): bench.py starts its own two ranks
    as child processes, runs BASELINE configs[3]'s flow (fixed global batch cut into contiguous shards, one gather per
    step, strong scaling) - rehearsed here on this box's one GPU over gloo at a reduced batch - relays rank 0's JSON line
    and exits with the children's code."""
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HCTR_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "10", "--width", "320"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                    # ONE JSON line on stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["config"]["global_lines"] == 10
    assert res["config"]["lines_per_gpu"] == 5 and res["multi_gpu"]["world_size_reported_by_backend"] == 2
    assert res["value"] > 0 and len(res["multi_gpu"]["per_rank"]["rows"]) == 2
    assert "starting 2 ranks" in r.stderr and "process group up, world size 2" in r.stderr
    # the launcher form the driver documents still works (ranks come from the environment, nothing is spawned)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29518", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "6", "--width", "160"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["config"]["lines_per_gpu"] == 3


def test_bench_single_gpu_line_has_every_record():
    """The driver's command at reduced size: ONE JSON line carrying the three precision modes, the host-bracket rate, the
    roofline and the configs[2] / configs[4] records (the full-size figures come from the driver's own run)."""
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--batch", "4",
                        "--width", "256", "--no-cpu-baseline", "--no-extra-configs"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    for key in ("value", "value_f16x3", "value_auto", "value_incl_h2d", "roofline", "flagged_lines_auto", "setup_s", "lib"):
        assert key in res, key
    assert res["n_gpus"] == 1 and res["roofline"]["launches_per_step_and_layer"] == 1 and res["value_auto"] > 0


def _c2_columns(model, imgs, chunk=8):
    """per-column argmax [n, W] of the engine's logits (read back in chunks of 8 lines = 0.47 GB)"""
    out = np.zeros((imgs.shape[0], imgs.shape[2]), np.int64)
    for s0 in range(0, imgs.shape[0], chunk):
        out[s0:s0 + chunk] = model(imgs[s0:s0 + chunk]).argmax(axis=2).T
    return out


def test_config2_all_64_lines_against_the_real_reference(engine, engine_x3, codec, synth):
    """BASELINE configs[1] at full size against the REAL reference's outputs for all 64 lines (tests/golden/c2_lines.*,
    made by make_golden_c2.py): per-column argmax and greedy text, both precision modes, FIXED floors (recorded from
    gpurun r2a: f16 1507 flips / 128 000 columns, none above a reference margin of 0.21, 2.4 % CER; f16x3 4 flips, all
    below a margin of 7.1e-5 - fp32-rounding territory - 60 of 64 lines exact, 4 character edits)."""
    with open(os.path.join(GOLDEN, "c2_lines.json"), encoding="utf-8") as f:
        meta = json.load(f)
    g = np.load(os.path.join(GOLDEN, "c2_lines.npz"))
    ref_arg, margin = g["argmax"].astype(np.int64), g["margin"]
    scale = float(np.abs(g["max"]).max())
    imgs = synth.make_line_images(64, 2000, meta["seed"])
    ref_text = meta["greedy"]

    # ---- f16 (the timed default): tolerance-qualified ------------------------------------------------
    arg = _c2_columns(engine, imgs)
    check_f16_argmax(arg, ref_arg, margin, scale, MIN_SAFE["c2"])
    flips = arg != ref_arg
    assert not flips[margin > 0.3].any()                      # tighter recorded fact: measured largest 0.21
    assert flips.sum() <= 2000
    text = codec.labels_to_text(engine.greedy(imgs))
    edits = sum(ctc_ref.edit_distance(a, b) for a, b in zip(text, ref_text))
    assert edits <= F16_CER_MAX * sum(len(t) for t in ref_text), edits

    # ---- f16x3: fp32-grade ------------------------------------------------------------------------------
    arg3 = _c2_columns(engine_x3, imgs)
    flips3 = arg3 != ref_arg
    assert not flips3[margin > 2e-4].any()                    # (2e-4 = 4.8e-6 of the logit scale)
    assert flips3.sum() <= 12
    assert (arg3 == g["second"].astype(np.int64))[flips3].all()          # a flip only ever swaps the reference's top two
    text3 = codec.labels_to_text(engine_x3.greedy(imgs))
    assert sum(a == b for a, b in zip(text3, ref_text)) >= 58
    assert sum(ctc_ref.edit_distance(a, b) for a, b in zip(text3, ref_text)) <= 8
    for b in range(64):                                       # every line without a sub-2e-4 near-tie is EXACT
        if (margin[b] > 2e-4).all():
            assert text3[b] == ref_text[b], b


def test_fused_beam_front_end_equals_stored_logits_path(pkg, engine, synth, state_dict):
    """The beam front end without stored logits (head GEMM twice with reducing epilogues, kernels.h ConvArgs) against
    the stored-logits kernels (row_topk / row_candidates on the engine's own logits): top-k classes, log-probs, blank
    log-prob and the p > 0.001 candidate lists must be IDENTICAL (same float32 arithmetic, ties by lower class)."""
    from importlib import import_module
    model_mod = import_module(pkg.__name__ + ".model")
    C = synth.DEFAULT_VOCAB + 2
    for seed, widths, k in ((5, [200, 200, 200], 10), (52, [131, 100, 64, 17], 10), (6, [90], 32), (7, [48, 33], 3)):
        imgs = synth.make_line_images(len(widths), max(widths), seed)
        B, W = imgs.shape[0], imgs.shape[2]
        fe = engine.beam_frontend(imgs, k=k, widths=widths, want_candidates=True)          # fused
        logits = engine(imgs, widths=widths)                                                # [W,B,C] on the host
        ref = model_mod.beam_frontend_call(engine._ctx, None, 0, 0, None, np.ascontiguousarray(logits), 0, B, W, C, k, True)
        assert np.array_equal(fe["topk_idx"], ref["topk_idx"]), (seed, k)
        assert np.array_equal(fe["topk_logp"], ref["topk_logp"]), (seed, k)
        assert np.array_equal(fe["blank_logp"], ref["blank_logp"])
        assert np.array_equal(fe["cand_off"], ref["cand_off"])
        n = int(fe["cand_off"][-1])
        assert n > 0 and np.array_equal(fe["cand_idx"][:n], ref["cand_idx"][:n])
        assert np.array_equal(fe["cand_logp"][:n], ref["cand_logp"][:n])
    # k beyond the fused path's limit is served by the stored-logits kernels
    imgs = synth.make_line_images(1, 40, 8)
    fe = engine.beam_frontend(imgs, k=40)
    assert fe["topk_idx"].shape == (40, 1, 40) and (np.diff(fe["topk_logp"], axis=2) <= 0).all()
    # near-uniform logits overflow the per-row lists: the pass is redone through the stored logits, same results
    sd = dict(state_dict)
    sd["linear.weight"] = np.zeros_like(state_dict["linear.weight"])
    sd["linear.bias"] = np.full_like(state_dict["linear.bias"], 0.25)
    flat = pkg.hctr_model(C).cuda(0)
    flat.load_state_dict(sd)
    fe = flat.beam_frontend(imgs, k=10, want_candidates=True)
    assert np.array_equal(fe["topk_idx"][0, 0], np.arange(10))                # all equal: lowest classes first
    assert np.allclose(fe["topk_logp"], -np.log(C), atol=1e-4) and int(fe["cand_off"][-1]) == 0


def test_trained_like_checkpoint_text_exact_in_both_modes(pkg, synth):
    """north_star's "decoded text exact" on BASELINE configs[1] (64 x 1x128x2000): with the trained-like checkpoint
    (synth.make_state_dict(head="trained"): logits peaky like a trained CTC model's, margins in
    tests/golden/c2_trained_lines.json) the greedy text of ALL 64 lines equals the REAL reference's (fp32 CPU) in the
    default f16 mode - the mode bench.py times - and in f16x3. Logit tolerances as everywhere: f16 0.01*scale + 0.05,
    f16x3 2e-4*scale; argmax identical on every column whose reference margin exceeds twice the mode's tolerance."""
    with open(os.path.join(GOLDEN, "c2_trained_lines.json"), encoding="utf-8") as f:
        meta = json.load(f)
    g = np.load(os.path.join(GOLDEN, "c2_trained_lines.npz"))
    ref_arg, margin = g["argmax"].astype(np.int64), g["margin"]
    scale = float(np.abs(g["max"]).max())
    C = synth.DEFAULT_VOCAB + 2
    sd = synth.make_state_dict(C, seed=0, head="trained")
    imgs = synth.make_font_lines(64, meta["width"], meta["seed"])
    cd = pkg.ctc_codec(synth.characters())
    for mode, tol in (("f16", f16_tol(scale)), ("f16x3", X3_RTOL * scale)):
        m = pkg.hctr_model(C, precision=mode).cuda(0)
        m.load_state_dict(sd)
        text = cd.labels_to_text(m.greedy(imgs))
        assert text == meta["greedy"], (mode, sum(a != b for a, b in zip(text, meta["greedy"])))
        arg = np.zeros_like(ref_arg)
        mx = np.zeros_like(g["max"])
        for s0 in range(0, 64, 8):
            lg = m(imgs[s0:s0 + 8])
            arg[s0:s0 + 8] = lg.argmax(axis=2).T
            mx[s0:s0 + 8] = lg.max(axis=2).T
        assert float(np.abs(mx - g["max"]).max()) <= tol, mode
        safe = margin > 2 * tol
        assert safe.mean() >= 0.999 and np.array_equal(arg[safe], ref_arg[safe]), mode
        assert (arg != ref_arg).sum() <= 8, mode
        del m


def test_c_abi_rccl_gather_world_1(pkg, engine, synth):
    """hctr_comm_* / hctr_gather_labels (the plain-C caller's collective, include/hctr_hip.h): communicator over RCCL
    with the one rank this box has, fed with hctr_greedy's own outputs; the gathered rows must equal the inputs and the
    pad rows must come back empty. (World size 2 needs two GPUs: the N > 1 logic is covered over gloo.)"""
    import ctypes
    lib = pkg.load_library()
    uid = ctypes.create_string_buffer(128)
    assert lib.hctr_comm_unique_id(uid) == 0, lib.hctr_comm_last_error()
    comm = ctypes.c_void_p()
    assert lib.hctr_comm_create(ctypes.byref(comm), uid, 0, 1, 0) == 0, lib.hctr_comm_last_error()
    try:
        imgs = synth.make_line_images(3, 120, 9)
        B, W = imgs.shape[0], imgs.shape[2]
        labels = np.zeros((B, W), np.int32)
        lengths = np.zeros((B,), np.int32)
        _l = pkg._lib
        _l.check(lib.hctr_greedy(engine._ctx, _l.ptr(imgs), _l.U8, 0, None, B, W, _l.ptr(labels), _l.ptr(lengths)), engine._ctx)
        cap, per = int(lengths.max()), 4                         # one pad row: lines_per_rank = ceil(n / world) may exceed n_local
        out_l = np.full((per, cap), -1, np.int32)
        out_n = np.full((per,), -1, np.int32)
        rc = lib.hctr_gather_labels(comm, _l.ptr(labels), _l.ptr(lengths), B, W, per, cap, _l.ptr(out_l), _l.ptr(out_n))
        assert rc == 0, lib.hctr_comm_last_error()
        assert out_n.tolist() == lengths.tolist() + [0]
        for b in range(B):
            assert np.array_equal(out_l[b, :lengths[b]], labels[b, :lengths[b]]) and not out_l[b, lengths[b]:].any()
        assert lib.hctr_gather_labels(comm, _l.ptr(labels), _l.ptr(lengths), B, W, per, cap - 1, _l.ptr(out_l), _l.ptr(out_n)) == -1
    finally:
        lib.hctr_comm_destroy(comm)


def test_config5_beam_strings_equal_the_real_reference_end_to_end(pkg, synth):
    """BASELINE configs[4] end to end against the REAL reference (its fp32 CPU forward followed by its own ctc_codec beam
    search, tests/golden/c5_beam_lines.json from make_golden_c5.py): full-width trained-like-checkpoint lines through
    the engine's f16 forward, the fused device front end (log-softmax, top-10, p > 0.001 lists) and the C++ host prefix
    search must give the same strings - cbs_full with the toy-bigram and the zero LM, and cbs_skip (whose in-place
    single-candidate updates make strings of ~380 characters on these lines: the reference's behaviour, reproduced)."""
    with open(os.path.join(GOLDEN, "c5_beam_lines.json"), encoding="utf-8") as f:
        gold = json.load(f)
    C = synth.DEFAULT_VOCAB + 2
    imgs = synth.make_font_lines(gold["lines"], gold["width"], gold["seed"])
    for mode in ("f16", "f16x3"):
        m = pkg.hctr_model(C, precision=mode).cuda(0)
        m.load_state_dict(synth.make_state_dict(C, seed=0, head="trained"))
        cd = pkg.ctc_codec(synth.characters()).attach(m)
        assert cd.labels_to_text(m.greedy(imgs)) == gold["greedy"], mode
        for tag, skip, lm in (("full_toy", False, pkg.ToyBigramLM()), ("full_zero", False, pkg.ZeroLM()),
                              ("skip_toy", True, pkg.ToyBigramLM())):
            cd.use_beam_search, cd.skip_search, cd.use_tfm_pred, cd.use_tfm_score = True, skip, False, False
            cd.lm_panelty, cd.len_bonus, cd.beam_size, cd.search_depth, cd.ngram = 0.8, 4.8, 10, 10, lm
            fe = m.beam_frontend(imgs, k=10, want_candidates=skip)
            assert cd.decode_frontend(fe) == gold[tag], (mode, tag)
        del cd, m


def test_shape_changes_reuse_one_arena(pkg, synth, state_dict):
    """Ragged workloads present a new (lines, width) with almost every batch. The workspace lives in ONE arena that is
    re-carved per shape with only the stored conv borders re-zeroed: results must not depend on which shapes ran before
    (stale interior / border data of a wider, narrower, larger or smaller previous shape), including when a later call
    needs a part the layout did not have yet (logits for hctr_forward_logits, the beam scratch)."""
    C = synth.DEFAULT_VOCAB + 2
    shapes = [(3, 200), (2, 333), (5, 64), (1, 1000), (4, 97), (2, 32), (3, 200)]
    imgs = {sh: synth.make_line_images(sh[0], sh[1], 70 + i) for i, sh in enumerate(shapes)}
    want = {}
    for sh in set(shapes):                                   # each shape on a fresh engine: nothing ran before
        m = pkg.hctr_model(C).cuda(0)
        m.load_state_dict(state_dict)
        want[sh] = (m.greedy(imgs[sh]), m(imgs[sh]) if sh[1] <= 200 else None)
        del m
    m = pkg.hctr_model(C).cuda(0)
    m.load_state_dict(state_dict)
    for rnd in range(2):
        for sh in shapes if rnd == 0 else shapes[::-1]:
            got = m.greedy(imgs[sh])
            assert all(np.array_equal(a, b) for a, b in zip(got, want[sh][0])), (rnd, sh)
            if want[sh][1] is not None and rnd == 1:         # second round: the logits part joins the layout mid-way
                assert np.array_equal(m(imgs[sh]), want[sh][1]), sh
    fe = m.beam_frontend(imgs[(2, 32)], k=10)                # and the beam scratch
    m2 = pkg.hctr_model(C).cuda(0)
    m2.load_state_dict(state_dict)
    fe2 = m2.beam_frontend(imgs[(2, 32)], k=10)
    assert np.array_equal(fe["topk_idx"], fe2["topk_idx"]) and np.array_equal(fe["topk_logp"], fe2["topk_logp"])
    assert all(np.array_equal(a, b) for a, b in zip(m.greedy(imgs[(1, 1000)]), want[(1, 1000)][0]))


def test_nan_logits_follow_numpy_argmax(pkg, engine, synth, state_dict):
    """np.argmax (utils/ctc_codec.py:75) treats a NaN as the maximum and returns the FIRST one; the device argmax
    kernels - over caller-supplied logits and fused into the head GEMM - do the same."""
    rng = np.random.default_rng(12)
    C = 9
    logits = rng.normal(size=(14, 2, C)).astype(np.float32)
    logits[3, 0, 4] = np.nan
    logits[3, 0, 6] = np.nan                    # two NaNs in a row: the first wins
    logits[7, 1, 0] = np.nan                    # NaN on the blank
    logits[9, 0, :] = -np.inf                   # all -inf: index 0
    logits[10, 1, 2] = np.inf
    cd = pkg.ctc_codec(codec_cases.vocab(C)).attach(engine)
    assert cd.decode(logits) == ctc_ref.CtcCodecRef(codec_cases.vocab(C)).decode(logits)
    sd = dict(state_dict)
    bias = state_dict["linear.bias"].copy()
    bias[5] = np.nan                            # every column's logit of class 5 is NaN
    sd["linear.bias"] = bias
    m = pkg.hctr_model(synth.DEFAULT_VOCAB + 2).cuda(0)
    m.load_state_dict(sd)
    imgs = synth.make_line_images(2, 80, 4)
    assert [lab.tolist() for lab in m.greedy(imgs)] == [[5], [5]]
    assert (np.nanargmax(np.where(np.isnan(m(imgs)), np.inf, m(imgs)), axis=2) == 5).all()


def test_config3_widths_equal_the_real_reference(pkg, synth):
    """BASELINE configs[2] (mixed widths 800/1600/2400/3200): trained-like-checkpoint lines of every bucket width, and one
    ragged batch of all four widths padded together (NormalizePAD replicate pad; the reference decodes the pad columns
    too), against the REAL reference's greedy strings (tests/golden/c3_lines.json): bucket lines exact in the default f16
    mode; the ragged batch exact in f16x3 (and in auto mode: tests/test_gpu_auto.py)."""
    with open(os.path.join(GOLDEN, "c3_lines.json"), encoding="utf-8") as f:
        gold = json.load(f)
    C = synth.DEFAULT_VOCAB + 2
    m = pkg.hctr_model(C).cuda(0)
    m.load_state_dict(synth.make_state_dict(C, seed=0, head="trained"))
    cd = pkg.ctc_codec(synth.characters())
    for bi, w in enumerate(gold["widths"]):
        imgs = synth.make_font_lines(2, w, gold["seed"], line_offset=bi * 128)
        assert cd.labels_to_text(m.greedy(imgs)) == gold["buckets"][str(w)], w
    widths = gold["ragged"]["widths"]
    batch = np.zeros((len(widths), 128, max(widths)), np.uint8)
    for i, w in enumerate(widths):
        batch[i, :, :w] = synth.make_font_lines(1, w, gold["seed"], line_offset=gold["widths"].index(w) * 128)[0]
    wd = np.array(widths, np.int32)
    # the replicate-pad region of a short line is constant input: the logits there are NOT peaky, and the f16 text differs
    # from the reference's on the short lines (gpurun_out/r2m). The mode that serves this batch shape is "auto"
    # (tests/test_gpu_auto.py::test_auto_mode_ragged_config3_batch_equals_the_real_reference asserts EQUALITY on all four
    # lines); plain f16 is only held to equality on the full-width line here, f16x3 on all four.
    got = cd.labels_to_text(m.greedy(batch, widths=wd))
    assert got[0] == gold["ragged"]["greedy"][0]
    m3 = pkg.hctr_model(C, precision="f16x3").cuda(0)
    m3.load_state_dict(synth.make_state_dict(C, seed=0, head="trained"))
    assert cd.labels_to_text(m3.greedy(batch, widths=wd)) == gold["ragged"]["greedy"]


def test_config3_full_size_bucketed_batch():
    """BASELINE configs[2] at FULL size: 512 lines in four equal-width buckets of 128 (widths 800/1600/2400/3200, the
    buckets beyond HCTR_MAX_COLS run in balanced internal passes). A line's labels must not depend on its batch: lines
    run alone give the labels they got inside their bucket; two runs are identical; once every bucket shape has been
    seen the workspace arena is neither re-allocated nor grown."""
    import importlib
    from conftest import PKG
    pkg = importlib.import_module(PKG)
    synth = pkg.synth
    C = synth.DEFAULT_VOCAB + 2
    m = pkg.hctr_model(C).cuda(0)
    m.load_state_dict(synth.make_state_dict(C, seed=0))
    buckets = [(w, synth.make_line_images(128, w, 3, line_offset=bi * 128)) for bi, w in enumerate((800, 1600, 2400, 3200))]
    first = [m.greedy(imgs) for _, imgs in buckets]
    st0 = m.workspace_stats()
    second = [m.greedy(imgs) for _, imgs in buckets]
    st1 = m.workspace_stats()
    assert st1["arena_allocations"] == st0["arena_allocations"] and st1["arena_bytes"] == st0["arena_bytes"]
    assert st0["arena_bytes"] < 16 * 2 ** 30              # (buckets of this size use the shared-buffer layout)
    for a, b in zip(first, second):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert sum(len(x) for x in first) == 512 and all(len(lab) > 0 for x in first for lab in x)
    for (w, imgs), labs in zip(buckets, first):
        assert m.lines_per_pass(128, w) == {800: 128, 1600: 64, 2400: 43, 3200: 32}[w]
        for i in (0, 42, 43, 127):                          # first / last lines of internal passes, run ALONE
            alone = m.greedy(imgs[i:i + 1])
            assert np.array_equal(alone[0], labs[i]), (w, i)


def test_config5_full_size_beam_batch():
    """BASELINE configs[4] at FULL size: 256 lines x 1x128x2000, cbs_full beam 10 / depth 10 with the toy-bigram LM.
    The pipelined decode (front end of chunk i+1 on the GPU while the host searches chunk i) must equal the two stages
    back to back on ALL 256 lines, and the six lines of tests/golden/c5_beam_lines.json (strings of the REAL reference's
    forward + codec) must come out exactly inside the big batch (lines 0-5; trained-like checkpoint)."""
    import importlib
    from conftest import PKG
    pkg = importlib.import_module(PKG)
    synth = pkg.synth
    pipe = importlib.import_module(PKG + ".pipeline")
    with open(os.path.join(GOLDEN, "c5_beam_lines.json"), encoding="utf-8") as f:
        gold = json.load(f)
    assert gold["width"] == 2000
    C = synth.DEFAULT_VOCAB + 2
    m = pkg.hctr_model(C).cuda(0)
    m.load_state_dict(synth.make_state_dict(C, seed=0, head="trained"))
    imgs = np.concatenate([synth.make_font_lines(gold["lines"], 2000, gold["seed"]),
                           synth.make_font_lines(256 - gold["lines"], 2000, 5)], axis=0)
    cd = pkg.ctc_codec(synth.characters()).attach(m)
    cd.use_beam_search, cd.skip_search, cd.use_tfm_pred, cd.use_tfm_score = True, False, False, False
    cd.lm_panelty, cd.len_bonus, cd.beam_size, cd.search_depth, cd.ngram = 0.8, 4.8, 10, 10, pkg.ToyBigramLM()
    seq = cd.decode_frontend(m.beam_frontend(imgs, k=10))
    assert len(seq) == 256 and seq[:gold["lines"]] == gold["full_toy"]
    for chunk in (32, 48):
        assert pipe.recognize_beam(m, cd, imgs, chunk=chunk) == seq, chunk
    # greedy text of the same batch through the fused path, for the six golden lines
    assert cd.labels_to_text(m.greedy(imgs[:gold["lines"]])) == gold["greedy"]


def test_aliased_workspace_layout_gives_identical_results(pkg, synth, state_dict):
    """HCTR_WS_ALIAS=1: the four stages share four activation buffers (a buffer changes geometry from stage to stage and
    run_forward re-zeroes the stored conv borders at every stage entry) instead of owning 15 dedicated ones. Same kernels,
    same data: labels and logits must be bit-identical to the dedicated layout's, over shape changes, unequal widths and
    all three precision modes; the arena of config 2 (64 x 2000) must stay below 13 GB (dedicated: 31 GB)."""
    C = synth.DEFAULT_VOCAB + 2
    ref = pkg.hctr_model(C, precision="auto").cuda(0)
    ref.load_state_dict(state_dict)
    os.environ["HCTR_WS_ALIAS"] = "1"                     # (read when the context is created)
    try:
        m = pkg.hctr_model(C, precision="auto").cuda(0)
        m.load_state_dict(state_dict)
    finally:
        os.environ.pop("HCTR_WS_ALIAS", None)
    cases = [(3, 200, None), (2, 333, [333, 120]), (5, 64, None), (1, 1000, None), (4, 97, [97, 64, 33, 5]), (3, 200, None)]
    for mode in ("f16", "f16x3", "auto"):
        ref.set_precision(mode)
        m.set_precision(mode)
        for i, (b, w, wd) in enumerate(cases):
            imgs = synth.make_line_images(b, w, 90 + i)
            wd = None if wd is None else np.array(wd, np.int32)
            a, r = m.greedy(imgs, widths=wd), ref.greedy(imgs, widths=wd)
            assert all(np.array_equal(x, y) for x, y in zip(a, r)), (mode, b, w)
            if w <= 200:
                assert np.array_equal(m(imgs, widths=wd), ref(imgs, widths=wd)), (mode, b, w)
    fe, fr = m.beam_frontend(imgs, k=10, want_candidates=True), ref.beam_frontend(imgs, k=10, want_candidates=True)
    assert all(np.array_equal(fe[k], fr[k]) for k in ("topk_idx", "topk_logp", "blank_logp", "cand_off"))
    with pytest.raises(RuntimeError):
        m.debug_activation("stage1", 3)                   # overwritten by later stages in this layout
    # full size (f16, fresh contexts without the optional logits / beam parts): the DEFAULT policy picks the shared
    # buffers by itself once the dedicated layout would pass 16 GiB (config 2: 31 GB) - identical labels to a context
    # forced to dedicated buffers, a third of the memory
    del m, ref
    os.environ["HCTR_WS_ALIAS"] = "0"
    try:
        ref = pkg.hctr_model(C).cuda(0)
        ref.load_state_dict(state_dict)
    finally:
        os.environ.pop("HCTR_WS_ALIAS", None)
    m = pkg.hctr_model(C).cuda(0)                           # default policy
    m.load_state_dict(state_dict)
    small = synth.make_line_images(3, 200, 5)
    m.greedy(small)
    assert m.debug_activation("stage1", 3).shape[0] == 3    # small shapes keep dedicated buffers: every tap readable
    big = synth.make_line_images(64, 2000, 2)
    a, r = m.greedy(big), ref.greedy(big)
    assert all(np.array_equal(x, y) for x, y in zip(a, r))
    sa, sr = m.workspace_stats(), ref.workspace_stats()
    assert sa["arena_bytes"] <= 13e9 < sr["arena_bytes"], (sa, sr)
    with pytest.raises(RuntimeError):
        m.debug_activation("stage1", 64)
