"""Width bucketing for variable-width line batches (BASELINE config 3).

The reference pads every batch to its widest line by replicating the last column
(``NormalizePAD``, utils/dataset.py:83-93, called at test.py:170-186) and the SE blocks average over
the padded width, so WHICH lines share a batch is part of the result. ``plan_batches`` therefore only
decides the grouping (similar widths together, to waste little padding); each batch is then run with
exactly the reference's semantics for that batch: common width = its maximum, per-line ``widths``.
"""
import numpy as np


def plan_batches(widths, max_lines, max_pad_fraction=0.1):
    """Group line indices into batches of at most ``max_lines`` lines whose widths differ by at most
    ``max_pad_fraction`` of the batch's widest line. Returns a list of int arrays (indices into the
    input order); widest batches first. Equal widths always end up together."""
    widths = np.asarray(widths, dtype=np.int64)
    order = np.argsort(-widths, kind="stable")
    batches, cur = [], []
    for i in order:
        if cur and (len(cur) >= max_lines or widths[i] < widths[cur[0]] * (1.0 - max_pad_fraction)):
            batches.append(np.array(cur, dtype=np.int64))
            cur = []
        cur.append(int(i))
    if cur:
        batches.append(np.array(cur, dtype=np.int64))
    return batches


def pad_batch(images, idx):
    """uint8 [n,128,maxW] + widths for the lines ``idx`` of a list of [128,w] arrays."""
    ws = np.array([images[i].shape[1] for i in idx], dtype=np.int32)
    out = np.zeros((len(idx), 128, int(ws.max())), dtype=np.uint8)
    for j, i in enumerate(idx):
        out[j, :, :ws[j]] = images[i]
    return out, ws


def recognize(model, codec, images, max_lines=64, max_pad_fraction=0.1):
    """Greedy-decode a list of uint8 [128,w] line images; returns strings in input order."""
    texts = [None] * len(images)
    for idx in plan_batches([im.shape[1] for im in images], max_lines, max_pad_fraction):
        batch, ws = pad_batch(images, idx)
        for i, lab in zip(idx, model.greedy(batch, widths=ws)):
            texts[i] = codec.labels_to_text([lab])[0]
    return texts
