"""Where config 5's pipelined beam decode spends its time (run on the GPU box):  python tools/gpu_c5_pipeline_times.py
Times, per 64-line chunk, the device front end (producer thread) and the host prefix search (consumer), run alone and
pipelined."""
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
model = hctr_amd.hctr_model(C).cuda(0)
model.load_state_dict(synth.make_state_dict(C, seed=0))
codec = hctr_amd.ctc_codec(synth.characters()).attach(model)
codec.use_beam_search, codec.use_tfm_pred, codec.skip_search = True, False, False
codec.beam_size = codec.search_depth = 10
codec.lm_panelty, codec.len_bonus, codec.ngram = 0.8, 4.8, hctr_amd.ToyBigramLM()
imgs = torch.from_numpy(synth.make_line_images(256, 2000, 5)).cuda()
torch.cuda.synchronize()
chunks = [imgs[i:i + 64] for i in range(0, 256, 64)]
model.beam_frontend(chunks[0], k=10)                      # warm
t = []
fes = []
for ch in chunks:
    t0 = time.perf_counter(); fes.append(model.beam_frontend(ch, k=10)); t.append(time.perf_counter() - t0)
print("front end alone, per chunk ms:", [round(x * 1e3, 1) for x in t])
t = []
for fe in fes:
    t0 = time.perf_counter(); codec.decode_frontend(fe); t.append(time.perf_counter() - t0)
print("host search alone, per chunk ms:", [round(x * 1e3, 1) for x in t])
# both at once on different chunks (what the pipeline does in steady state)
res = {}
def fe_loop():
    t0 = time.perf_counter()
    for ch in chunks: model.beam_frontend(ch, k=10)
    res["fe"] = (time.perf_counter() - t0) / len(chunks)
def hs_loop():
    t0 = time.perf_counter()
    for fe in fes: codec.decode_frontend(fe)
    res["hs"] = (time.perf_counter() - t0) / len(fes)
a, b = threading.Thread(target=fe_loop), threading.Thread(target=hs_loop)
t0 = time.perf_counter(); a.start(); b.start(); a.join(); b.join()
print("concurrently: front end %.1f ms/chunk, host search %.1f ms/chunk, wall %.1f ms for 4+4 chunks"
      % (res["fe"] * 1e3, res["hs"] * 1e3, (time.perf_counter() - t0) * 1e3))
for nt in (16, 32, 64, 128):
    codec.num_threads = nt
    t0 = time.perf_counter(); codec.decode_frontend(fes[0]); print("host search, %d threads: %.1f ms" % (nt, (time.perf_counter() - t0) * 1e3))

# the real pipeline (pipeline.recognize_beam), per-chunk wall times of both stages
import importlib  # noqa: E402
pipe = importlib.import_module(hctr_amd.package.__name__ + ".pipeline")
codec.num_threads = 0
for chunk, taper in ((64, True), (64, False), (32, True), (32, False)):
    pipe.recognize_beam(model, codec, imgs, chunk=chunk, taper=taper)          # warm (workspace shapes)
    st = {}
    t0 = time.perf_counter()
    pipe.recognize_beam(model, codec, imgs, chunk=chunk, taper=taper, stats=st)
    dt = time.perf_counter() - t0
    print("pipeline chunk=%d taper=%s: %.1f ms (%.1f lines/s)  chunks %s\n  front end ms %s (sum %.1f)\n  search ms %s\n  consumer wait ms %s"
          % (chunk, taper, dt * 1e3, 256 / dt, st["chunks"], st["frontend_ms"], sum(st["frontend_ms"]), st["search_ms"], st["consumer_wait_ms"]))
t0 = time.perf_counter(); fe = model.beam_frontend(imgs, k=10); print("one front-end call, 256 lines: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
