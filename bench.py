"""bench.py - hot-path throughput of the hctr engine on MI355X (driver contract: see DESIGN.md).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (NormalizePAD -> hctr trunk -> head -> argmax -> CTC collapse ->
labels on the host, + one RCCL gather to rank 0 when N > 1) over one batch of synthetic line images
per GPU: BASELINE.json configs[1], B=64 lines of 1x128x2000 per GPU, greedy decode. Inputs are
uint8 images already resident in HBM when the timed region starts. Weak scaling: per-GPU work is
fixed, value = all ranks' lines / max-over-ranks time.

Prints ONE JSON line on rank 0 with the `roofline` (dominant kernel, HIP events on the engine's own
stream during the timed steps) and `cpu_baseline` (the oracle restatement on the host cores, bounded
sample) objects.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W_LINE = 2000
B_PER_GPU = 64
SEED = 2
FLOP_PER_COL_DOM = 75.497472e6        # one 3x3 512->512 @ H=16 layer, per pixel column
FLOP_PER_COL_ALL = 1358.838912e6      # whole forward per pixel column at C=7358
PEAK_F16_TFLOPS = 2500.0              # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md)
# the ten 3x3 512->512 @H=16 launches of a forward (conv2 is named "+se" when the SE apply is fused)
DOMINANT = tuple(n + sfx for n in ("block3.0.conv2", "block3.1.conv2", "block3.2.conv2", "block3.3.conv2",
                                   "block3.4.conv2") for sfx in ("", "+se")) + \
    ("block3.1.conv1", "block3.2.conv1", "block3.3.conv1", "block3.4.conv1", "conv3+pool")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="lines per GPU")
    ap.add_argument("--width", type=int, default=W_LINE)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-lines", type=int, default=2)
    ap.add_argument("--layers", action="store_true", help="print the per-layer device-time table to stderr")
    ap.add_argument("--layer-file", default="", help="write the per-step launch order (layer names) to this file")
    ap.add_argument("--precision", default="f16", choices=["f16", "f16x3"],
                    help="engine precision mode for the timed run (default f16; f16x3 = split hi+lo pairs)")
    ap.add_argument("--no-pipeline", action="store_true", help="c5: run front end and host search back to back")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c5"],
                    help="c2 (default, the driver's line): B=64 x W=2000 greedy; c3: B=512 mixed widths "
                         "{800,1600,2400,3200} bucketed; c5: B=256 x W=2000 beam search 10/10 (extra modes, 1 GPU)")
    args = ap.parse_args()

    import torch
    import hctr_amd
    synth = hctr_amd.synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    dist = None
    # Test hook (rehearsing the N > 1 flow on a 1-GPU box): HCTR_BENCH_BACKEND=gloo shares the visible
    # GPUs round-robin between ranks and gathers over gloo. The driver's runs use RCCL ("nccl").
    backend = os.environ.get("HCTR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    dev = torch.device("cuda", local)

    C = synth.DEFAULT_VOCAB + 2
    B, W = args.batch, args.width
    sd = synth.make_state_dict(C, seed=0)
    model = hctr_amd.hctr_model(C, precision=args.precision).cuda(local)
    model.load_state_dict(sd)
    model.eval()
    if args.config != "c2":
        return extra_config(args, hctr_amd, model, sd, dev)
    # this rank's contiguous shard of the global synthetic batch (seed, global line index)
    imgs_host = synth.make_line_images(B, W, SEED, line_offset=rank * B)
    imgs = torch.from_numpy(imgs_host).to(dev)              # resident in HBM before timing
    torch.cuda.synchronize(dev)
    n_global = B * world
    import importlib
    gather = importlib.import_module(hctr_amd.package.__name__ + ".dist").gather_labels

    def step():
        labels = model.greedy(imgs)
        if dist is not None:
            return gather(labels, n_global, W, device=dev if backend == "nccl" else None)
        return labels

    for _ in range(args.warmup):
        out = step()
    model.set_profiling(True)
    prof = {}
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
        for name, ms in model.last_profile():
            prof.setdefault(name, []).append(ms)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    model.set_profiling(False)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    lines_per_s = n_global * args.steps / dt
    cols = B * W
    dom_ms = [np.mean(prof[n]) for n in DOMINANT if n in prof]
    dom_avg_ms = float(np.mean(dom_ms)) if dom_ms else float("nan")
    dom_tflops = FLOP_PER_COL_DOM * cols / (dom_avg_ms * 1e-3) / 1e12
    kernel_ms = float(sum(np.mean(v) for v in prof.values()))
    # HBM-side traffic of the dominant kernel per launch: rocprofv3 PMC passes of this same command
    # (tools/profile_rocprof.sh: FETCH_SIZE and WRITE_SIZE in separate runs, KiB units, FETCH_SIZE x2 on
    # gfx950 as MI355X_MICROARCH.md prescribes), summarised into profiles/per_layer_latest.json.
    traffic, traffic_src = None, None
    pj = os.path.join(ROOT, "profiles", "per_layer_latest.json")
    if os.path.isfile(pj) and B == B_PER_GPU and W == W_LINE and args.precision == "f16":   # (PMC passes ran in f16)
        with open(pj) as f:
            rows = [r for r in json.load(f) if r["layer"] in DOMINANT and r.get("fetch_gb_x2") is not None]
        if rows:
            traffic = float(np.mean([(r["fetch_gb_x2"] + r["write_gb"]) * 1e9 for r in rows]))
            traffic_src = "profiles/per_layer_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
    if args.layers:
        for name, v in prof.items():
            print("%-24s %9.3f ms" % (name, float(np.mean(v))), file=sys.stderr)
    if args.layer_file:
        with open(args.layer_file, "w") as f:
            f.write("\n".join(prof.keys()) + "\n")
    result = {
        "metric": "text-lines/sec (1x128x2000 synth) greedy decode",
        "value": round(lines_per_s, 3), "unit": "lines/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 (f32 accumulate)" if args.precision == "f16" else "f16x3 (hi+lo fp16 pairs, f32 accumulate)",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: B=%d x 1x128x%d uint8 lines per GPU, random-init hctr "
                               "(C=%d), forward + greedy CTC decode, labels to host" % (B, W, C),
                   "lines_per_gpu": B, "width": W, "classes": C, "parallelism": "batch-shard x%d" % world},
        "roofline": {"bound": "mfma", "kernel": "conv3x3_halo4 3x3 512->512 @H=16 (%d plain launches/step averaged)" % len(dom_ms),
                     "achieved": round(dom_tflops, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(dom_tflops / PEAK_F16_TFLOPS, 4), "traffic": traffic,
                     "traffic_unit": "bytes/launch (algorithmic: 2.36e9 in + 2.36e9 out + 4.7e6 weights)",
                     "traffic_source": traffic_src,
                     "avg_launch_ms": round(dom_avg_ms, 4),
                     "flops_per_launch": FLOP_PER_COL_DOM * cols},
        "whole_forward": {"kernel_ms_per_step": round(kernel_ms, 3),
                          "tflops": round(FLOP_PER_COL_ALL * cols / (kernel_ms * 1e-3) / 1e12, 2)},
    }

    if not args.no_cpu_baseline and world == 1:          # reported baseline: rank 0 of the 1-GPU run only
        from oracle import ctc_ref, hctr_ref
        nl = max(1, args.cpu_lines)
        x = synth.normalize_pad(imgs_host[:nl])
        codec = ctc_ref.CtcCodecRef(synth.characters())
        hctr_ref.forward(sd, x[:1, :, :, :64])                       # warm-up
        t0 = time.perf_counter()
        ref = hctr_ref.forward(sd, x).numpy()
        ref_txt = codec.decode(ref)
        cdt = time.perf_counter() - t0
        got_txt = [codec.characters and "".join(codec.characters[i] for i in lab) for lab in out[:nl]]
        ed = sum(ctc_ref.edit_distance(a, b) for a, b in zip(got_txt, ref_txt))
        result["cpu_baseline"] = {"value": round(nl / cdt, 4), "unit": "lines/s", "cores": torch.get_num_threads(),
                                  "kind": "port", "sample": "%d line(s) of 1x128x%d, oracle forward + greedy "
                                  "(torch CPU fp32)" % (nl, W)}
        result["parity_vs_cpu"] = {"mode": args.precision, "lines": nl,
                                   "exact_lines": int(sum(a == b for a, b in zip(got_txt, ref_txt))),
                                   "char_edits": int(ed), "ref_chars": int(sum(len(s) for s in ref_txt)),
                                   "note": "random-weight logits have many near-ties; f16 matches the reference's "
                                           "own TF32-class GPU precision, f16x3 is the fp32-grade mode"}
        if args.precision == "f16":          # the same lines through the split-precision mode (untimed)
            mx = hctr_amd.hctr_model(C, precision="f16x3").cuda(local)
            mx.load_state_dict(sd)
            lx = mx(imgs_host[:nl])
            x3_txt = ["".join(codec.characters[i] for i in lab) for lab in mx.greedy(imgs_host[:nl])]
            result["parity_vs_cpu_f16x3"] = {
                "lines": nl, "exact_lines": int(sum(a == b for a, b in zip(x3_txt, ref_txt))),
                "char_edits": int(sum(ctc_ref.edit_distance(a, b) for a, b in zip(x3_txt, ref_txt))),
                "max_logit_err": float(np.abs(lx - ref).max()), "logit_scale": float(np.abs(ref).max())}
            del mx
    print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


def extra_config(args, hctr_amd, model, sd, dev):
    """BASELINE configs 3 and 5 on one GPU (not the driver's line; results recorded in DESIGN.md)."""
    import torch
    synth = hctr_amd.synth
    C = synth.DEFAULT_VOCAB + 2
    codec = hctr_amd.ctc_codec(synth.characters()).attach(model)
    if args.config == "c3":
        buckets = []
        for bi, w in enumerate((800, 1600, 2400, 3200)):
            host = synth.make_line_images(128, w, 3, line_offset=bi * 128)
            buckets.append((w, host, torch.from_numpy(host).to(dev)))
        torch.cuda.synchronize(dev)

        def step():
            return [model.greedy(t) for _, _, t in buckets]
        n_lines, cols = 512, sum(128 * w for w, _, _ in buckets)
        name = "BASELINE configs[2]: B=512 mixed widths {800,1600,2400,3200}, 4 equal-width buckets of 128, greedy"
    else:
        host = synth.make_line_images(256, W_LINE, 5)
        dev_imgs = torch.from_numpy(host).to(dev)
        codec.use_beam_search, codec.use_tfm_pred, codec.skip_search = True, False, False
        codec.beam_size = codec.search_depth = 10
        codec.lm_panelty, codec.len_bonus = 0.8, 4.8          # test.py:74-79 defaults
        codec.ngram = hctr_amd.ToyBigramLM()
        t_front = [0.0]

        import importlib
        pipe = importlib.import_module(hctr_amd.package.__name__ + ".pipeline")

        def step():
            if args.no_pipeline:
                t0 = time.perf_counter()
                fe = model.beam_frontend(dev_imgs, k=10)
                t_front[0] += time.perf_counter() - t0
                return codec.decode_frontend(fe)
            return pipe.recognize_beam(model, codec, dev_imgs, chunk=64)
        n_lines, cols = 256, 256 * W_LINE
        name = ("BASELINE configs[4]: B=256 x 1x128x2000, cbs_full beam 10 / depth 10, toy-bigram LM, "
                "device log-softmax+top-k, C++ host prefix search on %d threads, %s" %
                (len(os.sched_getaffinity(0)), "sequential" if args.no_pipeline else "GPU front end pipelined with host search"))
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    res = {"metric": "text-lines/sec", "value": round(n_lines * args.steps / dt, 3), "unit": "lines/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
           "higher_is_better": True, "dtype": "f16 (f32 accumulate)", "data": "synthetic",
           "config": {"workload": name, "columns_per_step": cols},
           "columns_per_s": round(cols * args.steps / dt, 1)}
    if args.config == "c5":
        from oracle import ctc_ref, hctr_ref
        res["frontend_ms_per_step"] = round(t_front[0] / (args.steps + args.warmup) * 1e3, 2)
        # parity: the oracle codec (CPU) on the ENGINE's logits for 2 lines must give the same strings
        logits = model(host[:2])
        oc = ctc_ref.CtcCodecRef(synth.characters())
        oc.use_beam_search, oc.use_tfm_pred = True, False
        oc.lm_panelty, oc.len_bonus, oc.ngram = 0.8, 4.8, ctc_ref.ToyBigramLM()
        t0 = time.perf_counter()
        want = oc.decode(logits)
        res["cpu_codec_lines_per_s"] = round(2 / (time.perf_counter() - t0), 4)
        res["beam_strings_equal_oracle_on_engine_logits"] = bool(want == out[:2])
    print(json.dumps(res))


if __name__ == "__main__":
    main()
