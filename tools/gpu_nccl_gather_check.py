"""RCCL smoke of the result gather with the world this box offers (one rank per visible GPU, normally 1):
init_process_group("nccl") + dist.gather_labels on device tensors + barrier. The multi-rank logic itself is
covered by the gloo world-size-2 CPU test; this checks that the RCCL code path of bench.py runs on the box."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import hctr_amd  # noqa: E402

if "RANK" not in os.environ:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group(backend="nccl", device_id=dev)
import importlib  # noqa: E402
d = importlib.import_module(hctr_amd.package.__name__ + ".dist")
n, cap = 7 * world, 33
lo, hi = d.shard_range(n, rank, world)
rng = np.random.default_rng(5)
all_lines = [rng.integers(1, 7000, rng.integers(0, cap + 1)).astype(np.int32) for _ in range(n)]
got = d.gather_labels(all_lines[lo:hi], n, cap, device=dev)
dist.barrier()
torch.cuda.synchronize(dev)
if rank == 0:
    assert len(got) == n and all(np.array_equal(a, b) for a, b in zip(got, all_lines)), "gather mismatch"
    print("ok nccl gather world=%d" % world)
dist.destroy_process_group()
