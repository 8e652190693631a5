"""Multi-GPU batch sharding for the hctr path: one process per GPU, contiguous line ranges per rank,
no activation exchange, and ONE gather of the decoded label sequences to rank 0.

The reference has no multi-device inference (test.py:143-148 is single device; NCCL appears only in
training DDP, main.py:226-237). ``torch.distributed`` is plumbing here: backend "nccl" is RCCL over
xGMI on the GPU box, "gloo" on CPU for tests. The message is tiny (<= 4 MB per rank), so a direct
gather is latency-bound and the xGMI link rate is irrelevant.
"""
import numpy as np


def shard_range(n_lines, rank, world):
    """Contiguous range [lo, hi) of lines owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n_lines, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_labels(label_lists, cap, strict=True):
    """[n][1 + cap] int32: column 0 = length, then the labels (zero padded). A sequence longer than ``cap`` raises;
    with ``strict=False`` its row gets the length -1 instead (so a rank can still take part in the collective and every
    reader of the row sees the error)."""
    out = np.zeros((len(label_lists), 1 + cap), dtype=np.int32)
    for i, lab in enumerate(label_lists):
        n = len(lab)
        if n > cap:
            if strict:
                raise ValueError("label sequence longer than cap")
            out[i, 0] = -1
            continue
        out[i, 0] = n
        out[i, 1:1 + n] = lab
    return out


def unpack_labels(packed):
    return [row[1:1 + row[0]].copy() for row in packed]


def gather_labels(label_lists, n_lines, cap, device=None, group=None):
    """Gather every rank's decoded lines to rank 0 in global line order with ONE collective.

    Each rank passes the label arrays of its ``shard_range`` lines. All ranks send a buffer padded
    to the largest shard (ceil(n_lines / world)) so the collective is a plain ``gather``.
    Returns the full list on rank 0, None elsewhere.

    ``cap`` = labels per line in the packed buffer. Callers pass the padded width W: a collapsed sequence can be as
    long as W (alternating labels), and random-init weights really emit ~W/2 labels per line (978 of 2000 on
    BASELINE's synthetic lines), so SURVEY 8e's "Lmax < 128" does not hold for this workload and no smaller constant
    is safe without a second collective to agree on it. The message stays small (4.1 MB per rank at 512 lines x 2000).
    A rank whose sequence exceeds ``cap`` still takes part in the gather (length -1 in that row) and raises afterwards;
    rank 0 raises on seeing the sentinel - nobody is left waiting in the collective."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    per = -(-n_lines // world)
    buf = np.zeros((per, 1 + cap), dtype=np.int32)
    mine = pack_labels(label_lists, cap, strict=False)
    buf[:mine.shape[0]] = mine
    overflow = bool((mine[:, 0] < 0).any())
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    if dist.get_backend(group) == "nccl" or rank == 0:
        outs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    else:
        outs = None
    dist.gather(t, outs, dst=0, group=group)
    if overflow:
        raise ValueError("label sequence longer than cap=%d on rank %d" % (cap, rank))
    if rank != 0:
        return None
    result = []
    for r in range(world):
        lo, hi = shard_range(n_lines, r, world)
        rows = outs[r].cpu().numpy()[:hi - lo]
        if (rows[:, 0] < 0).any():
            raise ValueError("rank %d reported a label sequence longer than cap=%d" % (r, cap))
        result.extend(unpack_labels(rows))
    return result


def recognize_sharded(model, images, widths=None, device=None, group=None):
    """Greedy-decode a batch that every rank can index, each rank taking its contiguous ``shard_range``.

    ``images`` is the GLOBALLY padded uint8 batch [n,128,maxW] (plus each line's valid width): a line's SE means see
    its pad columns (models/handwritten_ctr_model.py:27 over NormalizePAD's replicate pad, utils/dataset.py:83-93), so
    the padded width must be fixed BEFORE sharding for N ranks to reproduce the one-rank result bit for bit
    (SURVEY.md 8e). Returns the n label arrays in input order on rank 0, None elsewhere."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n, max_w = int(images.shape[0]), int(images.shape[2])
    lo, hi = shard_range(n, rank, world)
    if hi > lo:
        mine = model.greedy(images[lo:hi], widths=None if widths is None else widths[lo:hi])
    else:
        mine = []
    return gather_labels(mine, n, max_w, device=device, group=group)
