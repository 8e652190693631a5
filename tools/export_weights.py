"""Write a state dict as the flat weights.bin that examples/greedy_demo.c reads (no torch needed by the C caller):

    python tools/export_weights.py synthetic[:seed] weights.bin          # the package's synthetic checkpoint
    python tools/export_weights.py hctr_checkpoint.pth.tar weights.bin   # a real checkpoint (weights_only load)

Per entry: u32 key length, key bytes, u32 dtype (1 = float32, 2 = int64), u32 ndim, i64 shape[ndim], raw data."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hctr_amd  # noqa: E402


def main(src, dst, num_classes=None):
    if src.startswith("synthetic"):
        seed = int(src.split(":")[1]) if ":" in src else 0
        sd = hctr_amd.synth.make_state_dict(num_classes or hctr_amd.synth.DEFAULT_VOCAB + 2, seed=seed)
    else:
        import torch
        ck = torch.load(src, map_location="cpu", weights_only=True)
        sd = {k: v.numpy() for k, v in (ck["state_dict"] if "state_dict" in ck else ck).items()}
    with open(dst, "wb") as f:
        for k, v in sd.items():
            a = np.ascontiguousarray(v)
            dt = 2 if a.dtype == np.int64 else 1
            if dt == 1:
                a = a.astype(np.float32, copy=False)
            kb = k.encode()
            f.write(struct.pack("<I", len(kb)) + kb + struct.pack("<II", dt, a.ndim))
            f.write(struct.pack("<%dq" % a.ndim, *a.shape))
            f.write(a.tobytes())
    print("%d tensors -> %s" % (len(sd), dst))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else None)
