"""Worker of test_gpu_parity.py::test_two_ranks_equal_one_rank (one process per rank, torch.distributed.run):
shards a 6-line unequal-width batch over the ranks (all sharing the box's one GPU, results gathered over gloo)
with the product's dist.recognize_sharded, and on rank 0 compares the gathered label arrays - values AND order -
with the same batch decoded by one process in one call."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist  # noqa: E402

import hctr_amd  # noqa: E402

hdist = importlib.import_module(hctr_amd.package.__name__ + ".dist")


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    synth = hctr_amd.synth
    C = synth.DEFAULT_VOCAB + 2
    widths = np.array([300, 211, 131, 64, 17, 280], dtype=np.int32)      # pad width matters for 5 of the 6 lines
    imgs = synth.make_line_images(len(widths), int(widths.max()), 61)
    model = hctr_amd.hctr_model(C).cuda(0)
    model.load_state_dict(synth.make_state_dict(C, seed=0))
    got = hdist.recognize_sharded(model, imgs, widths)
    if rank == 0:
        want = model.greedy(imgs, widths=widths)                           # one rank, one call
        assert len(got) == len(want) == len(widths)
        assert all(np.array_equal(a, b) for a, b in zip(got, want)), "sharded result differs from the one-rank result"
        assert sum(len(x) for x in want) > 20
        # a shard decoded WITHOUT the global pad width is a different computation: the test would notice
        lo, hi = hdist.shard_range(len(widths), world - 1, world)
        w_local = int(widths[lo:hi].max())
        if w_local < int(widths.max()):
            local = model.greedy(np.ascontiguousarray(imgs[lo:hi, :, :w_local]), widths=widths[lo:hi])
            print("LOCALLY_PADDED_DIFFERS", any(not np.array_equal(a, b) for a, b in zip(local, want[lo:hi])))
        print("IDENTITY_OK world=%d lines=%d" % (world, len(want)))
    else:
        assert got is None
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
