"""Child process of test_host_cpu.py::test_abi_no_exception_crosses: runs the pure-host beam search
(hctr_beam_search needs no GPU) under a lowered RLIMIT_AS so that thread creation and/or heap growth fail
inside the library, and prints the status it returned. A C++ exception escaping an extern "C" body would
abort this process instead (std::terminate)."""
import ctypes
import importlib
import os
import resource
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_lib = importlib.import_module("handwritten-chinese-ocr-samples_amd._lib")


def vm_bytes():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[0]) * os.sysconf("SC_PAGE_SIZE")


def main():
    headroom_mb, threads = int(sys.argv[1]), int(sys.argv[2])
    lib = _lib.load()
    W, B, C, k = 6000, 8, 400, 10
    rng = np.random.default_rng(5)
    # flat log-probs: every step extends every beam with every candidate, so each line's prefix trie keeps growing
    topk_idx = np.ascontiguousarray(np.argsort(rng.random((W, B, C)), axis=2)[:, :, :k].astype(np.int32))
    topk_idx[:, :, 0] = 1 + (np.arange(W)[:, None] % 300)          # greedy line is never empty
    topk_logp = np.full((W, B, k), np.log(1.0 / k), dtype=np.float32)
    blank = np.full((W, B), -8.0, dtype=np.float32)
    labels = np.zeros((B, W), dtype=np.int32)
    lengths = np.zeros((B,), dtype=np.int32)
    status = np.zeros((B,), dtype=np.int32)
    p = _lib.BeamParams()
    p.skip_search, p.beam_size, p.search_depth = 0, 10, 10
    p.lm_panelty, p.len_bonus, p.builtin_lm, p.num_threads = 0.8, 4.8, 1, threads
    soft, hard = resource.getrlimit(resource.RLIMIT_AS)
    resource.setrlimit(resource.RLIMIT_AS, (vm_bytes() + headroom_mb * (1 << 20), hard))
    rc = lib.hctr_beam_search(ctypes.byref(p), W, B, C, k, _lib.ptr(topk_idx), _lib.ptr(topk_logp), _lib.ptr(blank),
                              None, None, None, None, _lib.ptr(labels), _lib.ptr(lengths), _lib.ptr(status))
    resource.setrlimit(resource.RLIMIT_AS, (soft, hard))
    print("RC %d LEN %d" % (rc, int(lengths.sum())))


if __name__ == "__main__":
    main()
