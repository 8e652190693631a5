"""Experiment (GPU box): does running TWO engine contexts (two HIP streams, two workspaces) concurrently on halves of a
batch fill the launch gaps / kernel tails of a single stream?   python tools/gpu_two_streams.py
Prints lines/s of config 2 (64 x 1x128x2000, greedy) for: one context x 64 lines; two contexts x 32 lines on two host
threads; two contexts x 64 lines (128 lines in flight)."""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hctr_amd  # noqa: E402

synth = hctr_amd.synth
C = synth.DEFAULT_VOCAB + 2
sd = synth.make_state_dict(C, seed=0)
models = []
for _ in range(2):
    m = hctr_amd.hctr_model(C).cuda(0)
    m.load_state_dict(sd)
    models.append(m)
imgs = torch.from_numpy(synth.make_line_images(128, 2000, 2)).cuda()
torch.cuda.synchronize()


def run(parts, steps=6, warm=2):
    """parts: list of (model, tensor) run concurrently, one host thread each; returns seconds per step"""
    def loop(m, x, n):
        for _ in range(n):
            m.greedy(x)
    for n in (warm, steps):
        th = [threading.Thread(target=loop, args=(m, x, n)) for m, x in parts]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    return dt


for rnd in range(2):
    a = run([(models[0], imgs[:64])])
    b = run([(models[0], imgs[:32]), (models[1], imgs[32:64])])
    c = run([(models[0], imgs[:64]), (models[1], imgs[64:128])])
    d = run([(models[0], imgs[:128])])
    print("round %d: 1 ctx x 64: %.1f lines/s | 2 ctx x 32: %.1f | 2 ctx x 64: %.1f | 1 ctx x 128: %.1f"
          % (rnd, 64 / a, 64 / b, 128 / c, 128 / d), flush=True)
ref = [x.tolist() for x in models[0].greedy(imgs[:64])]
got = [x.tolist() for x in models[0].greedy(imgs[:32])] + [x.tolist() for x in models[1].greedy(imgs[32:64])]
print("labels identical:", ref == got)
