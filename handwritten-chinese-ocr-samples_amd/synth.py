"""Deterministic synthetic weights, vocabulary and line images for the hctr path.

Nothing here depends on torch's RNG stream: every value is a pure function of
``(seed, tensor-key, element-index)`` through a splitmix64 hash, so this
container, the GPU box and any later round regenerate bit-identical tensors
(the 212 MB fp32 state dict cannot be committed).

The state dict follows the reference's checkpoint schema exactly
(``models/handwritten_ctr_model.py:63-169`` as serialised by ``main.py:349-356``):
254 entries, conv weights ``[Cout, Cin, kh, kw]``, BN with running stats and an
int64 ``num_batches_tracked``, SE ``fc.0/fc.2`` without bias, ``linear`` ``[C, 2048]``.

"Diversified" init: BN statistics are randomised so
a BN-folding bug is visible, and the head is scaled so that the argmax varies
across columns instead of being bias-dominated.
"""
import os
import zlib

import numpy as np

IMG_H = 128
FEAT = 2048
DEFAULT_VOCAB = 7356          # synthetic vocabulary size (C = V + 2 = 7358)
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays."""
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hash_u64(seed, stream, n, offset=0):
    """n 64-bit hashes for (seed, stream, offset..offset+n)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([(int(seed) * 0x100000001B3 + int(stream)) & 0xFFFFFFFFFFFFFFFF],
                                    dtype=np.uint64))[0]
        idx = np.arange(offset, offset + n, dtype=np.uint64)
        return _splitmix64(base ^ (idx * np.uint64(0xD6E8FEB86659FD93) & _M64))


def uniform01(seed, stream, n, offset=0):
    """float32 uniform in [0,1) with 24 random bits (exactly representable)."""
    h = hash_u64(seed, stream, n, offset)
    return ((h >> np.uint64(40)).astype(np.float32)) * np.float32(1.0 / (1 << 24))


def key_stream(key):
    return zlib.crc32(key.encode("utf-8")) & 0xFFFFFFFF


def _u(seed, key, shape, lo, hi):
    n = int(np.prod(shape))
    u = uniform01(seed, key_stream(key), n)
    return (np.float32(lo) + u * np.float32(hi - lo)).reshape(shape).astype(np.float32)


def characters(vocab=DEFAULT_VOCAB):
    """Synthetic vocabulary: consecutive CJK code points from U+4E00."""
    return "".join(chr(0x4E00 + i) for i in range(vocab))


# ---------------------------------------------------------------------------
# network description shared by the generator, the oracle and the engine tests
# ---------------------------------------------------------------------------
STAGE_PLANES = [128, 256, 512, 512]   # block1..block4 (models/handwritten_ctr_model.py:66-70)
STAGE_BLOCKS = [2, 4, 5, 1]           # hctr_model: ResNet(1, 512, BasicBlock, [2,4,5,1]) (:166)


def conv_specs():
    """(conv_key, bn_key, cin, cout, ksize, has_bias) in forward order."""
    out = [("cnn.conv0_1", "cnn.bn0_1", 1, 64, 3, True),
           ("cnn.conv0_2", "cnn.bn0_2", 64, 64, 3, True)]
    inpl = 64
    for s, (planes, nb) in enumerate(zip(STAGE_PLANES, STAGE_BLOCKS), start=1):
        for i in range(nb):
            p = "cnn.block%d.%d" % (s, i)
            if i == 0 and inpl != planes:
                out.append((p + ".downsample.0", p + ".downsample.1", inpl, planes, 1, False))
            out.append((p + ".conv1", p + ".bn1", inpl, planes, 3, True))
            out.append((p + ".conv2", p + ".bn2", planes, planes, 3, True))
            inpl = planes
        out.append(("cnn.conv%d" % s, "cnn.bn%d" % s, planes, planes, 3, True))
    return out


def se_specs():
    out = []
    for s, (planes, nb) in enumerate(zip(STAGE_PLANES, STAGE_BLOCKS), start=1):
        for i in range(nb):
            out.append(("cnn.block%d.%d.se" % (s, i), planes))
    return out


_CALIB_CACHE = {}


def _load_calib():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_bn_calib.npz")
    if "v" not in _CALIB_CACHE:
        if not os.path.isfile(path):
            raise FileNotFoundError(path + " missing: run tests/golden/calibrate_synth_bn.py")
        with np.load(path, allow_pickle=False) as z:
            _CALIB_CACHE["v"] = {k: z[k] for k in z.files}
    return _CALIB_CACHE["v"]


def make_state_dict(num_classes=DEFAULT_VOCAB + 2, seed=0, diversified=True, calib="auto", head="random"):
    """Full 254-entry state dict (numpy arrays, reference key schema).

    ``calib="auto"`` loads the committed trained-like BN statistics
    (``synth_bn_calib.npz``, produced by tests/golden/calibrate_synth_bn.py for seed 0)
    and perturbs them per seed; ``calib=None`` draws arbitrary statistics instead.

    ``head="random"`` (default): heavy-tailed random classifier - its logits have a near-tie in about one
    column of eight, which no reduced-precision path can decode identically to fp32. ``head="trained"``: the
    classifier rows of the glyph-font classes (``make_font_lines``) come from ``synth_head_trained.npz``, a
    ridge-regression read-out fitted on this same trunk's features (tools/fit_trained_head.py), so the logits on
    font lines are peaky like a trained CTC model's; seed 0 / the default vocabulary only.
    """
    cal = _load_calib() if (calib == "auto" and diversified) else None
    sd = {}
    for ck, bk, cin, cout, ks, has_bias in conv_specs():
        fan_in = cin * ks * ks
        a = np.sqrt(6.0 / fan_in)                 # He-uniform: keeps post-ReLU scale O(1)
        sd[ck + ".weight"] = _u(seed, ck + ".weight", (cout, cin, ks, ks), -a, a)
        if has_bias:
            sd[ck + ".bias"] = _u(seed, ck + ".bias", (cout,), -0.1, 0.1)
        if diversified:
            sd[bk + ".weight"] = _u(seed, bk + ".weight", (cout,), 0.7, 1.3)
            sd[bk + ".bias"] = _u(seed, bk + ".bias", (cout,), -0.2, 0.2)
            if cal is not None:
                var = cal[bk + ".running_var"] * _u(seed, bk + ".running_var", (cout,), 0.8, 1.25)
                mean = cal[bk + ".running_mean"] + \
                    np.sqrt(var) * _u(seed, bk + ".running_mean", (cout,), -0.1, 0.1)
                sd[bk + ".running_mean"] = mean.astype(np.float32)
                sd[bk + ".running_var"] = var.astype(np.float32)
            else:
                sd[bk + ".running_mean"] = _u(seed, bk + ".running_mean", (cout,), -0.2, 0.2)
                sd[bk + ".running_var"] = _u(seed, bk + ".running_var", (cout,), 0.6, 1.6)
        else:
            sd[bk + ".weight"] = np.ones((cout,), np.float32)
            sd[bk + ".bias"] = np.zeros((cout,), np.float32)
            sd[bk + ".running_mean"] = np.zeros((cout,), np.float32)
            sd[bk + ".running_var"] = np.ones((cout,), np.float32)
        sd[bk + ".num_batches_tracked"] = np.array(1000, dtype=np.int64)
    for sk, c in se_specs():
        r = c // 16
        a0 = 2.0 * np.sqrt(3.0 / c)
        a2 = 2.0 * np.sqrt(3.0 / r)
        sd[sk + ".fc.0.weight"] = _u(seed, sk + ".fc.0.weight", (r, c), -a0, a0)
        sd[sk + ".fc.2.weight"] = _u(seed, sk + ".fc.2.weight", (c, r), -a2, a2)
    # head: the signed cube of a uniform is heavy-tailed, so a column's logits have a clear
    # winner more often than Gaussian logits would; the bias removes the response to the
    # mean feature vector so the argmax follows the image content, and lifts the blank.
    u = _u(seed, "linear.weight", (num_classes, FEAT), -1.0, 1.0)
    w = (u * u * u * np.float32(HEAD_SCALE)).astype(np.float32)
    b = _u(seed, "linear.bias", (num_classes,), -0.5, 0.5)
    if cal is not None:
        b = (b.astype(np.float64) - w.astype(np.float64) @ cal["feat_mean"].astype(np.float64))
        b = b.astype(np.float32)
    b[0] += np.float32(HEAD_BLANK_BIAS)
    if head == "trained":
        if seed != 0 or num_classes != DEFAULT_VOCAB + 2 or cal is None:
            raise ValueError("the trained-like head exists for seed 0 and the default vocabulary only")
        th = _load_trained_head()
        # classes outside the font: small random rows far below the fitted ones (they never win a column)
        w = (w * np.float32(0.05)).astype(np.float32)
        b = (_u(seed, "linear.bias", (num_classes,), -0.5, 0.5) - np.float32(8.0)).astype(np.float32)
        rows = np.array([0] + [font_label(k) for k in range(th["w"].shape[0] - 1)], dtype=np.int64)   # row 0 = <blank>
        w[rows] = th["w"].astype(np.float32)
        b[rows] = th["b"].astype(np.float32)
    elif head != "random":
        raise ValueError("head must be 'random' or 'trained'")
    sd["linear.weight"] = w
    sd["linear.bias"] = b.astype(np.float32)
    return sd


def _load_trained_head():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_head_trained.npz")
    if "head" not in _CALIB_CACHE:
        if not os.path.isfile(path):
            raise FileNotFoundError(path + " missing: run tools/fit_trained_head.py")
        with np.load(path, allow_pickle=False) as z:
            _CALIB_CACHE["head"] = {k: z[k] for k in z.files}
    return _CALIB_CACHE["head"]


HEAD_SCALE = 0.35
HEAD_BLANK_BIAS = 15.5


def to_torch(sd):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


# ---------------------------------------------------------------------------
# synthetic line images (uint8 [B,128,W]; background 255, dark strokes)
# ---------------------------------------------------------------------------
def make_line_images(batch, width, seed, line_offset=0):
    """Stroke-like synthetic text lines. Deterministic per (seed, line)."""
    imgs = np.full((batch, IMG_H, width), 255, dtype=np.int16)
    yy = np.arange(IMG_H, dtype=np.float32)[:, None]
    for b in range(batch):
        line = line_offset + b
        st = (int(seed) << 20) ^ line
        r = uniform01(st, 1, 4096)
        ri = 0
        x = 4 + int(r[ri] * 20); ri += 1
        while x < width - 8:
            gw = 40 + int(r[ri] * 50); ri += 1
            nseg = 6 + int(r[ri] * 9); ri += 1
            if ri + nseg * 6 + 8 >= r.size:
                r = uniform01(st, 2 + x, 4096); ri = 0
            x0g, x1g = x, min(width, x + gw)
            xx = np.arange(x0g, x1g, dtype=np.float32)[None, :]
            for _ in range(nseg):
                ax = x0g + r[ri] * (x1g - x0g); ay = 16 + r[ri + 1] * 96
                bx = x0g + r[ri + 2] * (x1g - x0g); by = 16 + r[ri + 3] * 96
                th = 1.0 + r[ri + 4] * 1.5
                val = int(r[ri + 5] * 96); ri += 6
                dx, dy = bx - ax, by - ay
                den = dx * dx + dy * dy + 1e-3
                t = np.clip(((xx - ax) * dx + (yy - ay) * dy) / den, 0.0, 1.0)
                d2 = (xx - ax - t * dx) ** 2 + (yy - ay - t * dy) ** 2
                m = d2 <= th * th
                sub = imgs[b, :, x0g:x1g]
                sub[m] = np.minimum(sub[m], val)
            x += gw + int(r[ri] * 12); ri += 1
        noise = (uniform01(st, 7, IMG_H * width) * 13.0).astype(np.int16).reshape(IMG_H, width) - 6
        imgs[b] += noise
    return np.clip(imgs, 0, 255).astype(np.uint8)


# ---------------------------------------------------------------------------
# glyph-font lines: the same stroke drawing, but every glyph is one of FONT_CLASSES fixed shapes, so a line
# has a ground-truth label string and a read-out fitted on trunk features can recognise it
# ---------------------------------------------------------------------------
FONT_CLASSES = 8              # what a linear read-out of the RANDOM trunk's 2048 features separates cleanly
FONT_SEED = 0x51F0            # (tools/fit_trained_head.py: 8 classes decode at 2.6 % CER vs the truth, 32 at 26 %)


def font_label(k):
    """label id (index into ``characters()`` + 1) of font class k: spread over the whole vocabulary / every head tile"""
    return 1 + (int(k) * 919 + 3) % DEFAULT_VOCAB


_FONT_CACHE = {}


def font_glyph(k):
    """(width, [(ax, ay, bx, by, thickness, value)]) of font class k; coordinates relative to the glyph box"""
    if k not in _FONT_CACHE:
        r = uniform01(FONT_SEED, 1000 + int(k), 128)
        gw = 40 + int(r[0] * 50)
        nseg = 6 + int(r[1] * 9)
        segs = []
        for i in range(nseg):
            q = r[2 + 6 * i:8 + 6 * i]
            segs.append((float(q[0]) * gw, 16 + float(q[1]) * 96, float(q[2]) * gw, 16 + float(q[3]) * 96,
                         1.0 + float(q[4]) * 1.5, int(q[5] * 96)))
        _FONT_CACHE[k] = (gw, segs)
    return _FONT_CACHE[k]


def make_font_lines(batch, width, seed, line_offset=0, with_truth=False, n_classes=FONT_CLASSES):
    """uint8 [B,128,W] lines made of font glyphs (deterministic per (seed, line)). With ``with_truth`` also returns,
    per line, the list of (class k, x0, x1) boxes in drawing order (glyphs cut by the right edge included)."""
    imgs = np.full((batch, IMG_H, width), 255, dtype=np.int16)
    yy = np.arange(IMG_H, dtype=np.float32)[:, None]
    truth = []
    for b in range(batch):
        line = line_offset + b
        st = ((int(seed) << 20) ^ line) + 0x7F000000
        r = uniform01(st, 1, 512)
        ri = 0
        x = 4 + int(r[ri] * 20); ri += 1
        boxes = []
        while x < width - 8:
            k = int(r[ri] * n_classes) % n_classes; ri += 1
            gw, segs = font_glyph(k)
            x0g, x1g = x, min(width, x + gw)
            xx = np.arange(x0g, x1g, dtype=np.float32)[None, :]
            for ax, ay, bx, by, th, val in segs:
                ax, bx = x0g + ax, x0g + bx
                dx, dy = bx - ax, by - ay
                den = dx * dx + dy * dy + 1e-3
                t = np.clip(((xx - ax) * dx + (yy - ay) * dy) / den, 0.0, 1.0)
                d2 = (xx - ax - t * dx) ** 2 + (yy - ay - t * dy) ** 2
                m = d2 <= th * th
                sub = imgs[b, :, x0g:x1g]
                sub[m] = np.minimum(sub[m], val)
            boxes.append((k, x0g, x0g + gw))
            x += gw + int(r[ri] * 12); ri += 1
        noise = (uniform01(st, 7, IMG_H * width) * 13.0).astype(np.int16).reshape(IMG_H, width) - 6
        imgs[b] += noise
        truth.append(boxes)
    out = np.clip(imgs, 0, 255).astype(np.uint8)
    return (out, truth) if with_truth else out


def font_truth_text(boxes, width, chars=None):
    """label string of a line: every glyph whose box lies completely inside the image"""
    chars = chars or characters()
    return "".join(chars[font_label(k) - 1] for k, x0, x1 in boxes if x1 <= width)


def normalize_pad(images_u8, widths=None, max_w=None):
    """NormalizePAD (utils/dataset.py:83-93): x/255 -> (x-0.5)/0.5, right pad by
    replicating the last valid column. ``images_u8`` is [B,128,W]; ``widths`` gives
    each line's valid width (default W). Returns float32 [B,1,128,max_w]."""
    b, h, w = images_u8.shape
    if widths is None:
        widths = [w] * b
    if max_w is None:
        max_w = max(int(x) for x in widths)
    out = np.zeros((b, 1, h, max_w), dtype=np.float32)
    for i in range(b):
        wi = int(widths[i])
        x = images_u8[i, :, :wi].astype(np.float32) / np.float32(255.0)
        x = (x - np.float32(0.5)) / np.float32(0.5)
        out[i, 0, :, :wi] = x
        if wi < max_w:
            out[i, 0, :, wi:] = x[:, wi - 1:wi]
    return out
