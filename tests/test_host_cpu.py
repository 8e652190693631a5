"""CPU tests (no GPU): the C ABI library loads and exports every symbol include/hctr_hip.h
declares, the product fails loudly without a GPU, and the host-side logic (C++ prefix beam search,
label packing, world-size-2 gather over gloo) matches the reference's golden outputs."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
from scipy.special import log_softmax

import codec_cases
from conftest import GOLDEN, ROOT


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    header = open(os.path.join(ROOT, "include", "hctr_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(hctr_[a-z_0-9]+)\s*\(", header))
    declared -= {"hctr_lm_score_cb", "hctr_lm_next_cb"}
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), name
    from importlib import import_module
    bound = {n for n, _, _ in import_module(pkg.__name__ + "._lib").SIGNATURES}
    assert bound == declared
    assert b"gfx950" in lib.hctr_version()


def test_no_gpu_fails_loudly(pkg):
    """Without a HIP device the engine raises; it never falls back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = pkg.hctr_model(100)
    with pytest.raises(RuntimeError):
        m.cuda(0)
    with pytest.raises(RuntimeError):
        m(np.zeros((1, 1, 128, 16), np.float32))
    cd = pkg.ctc_codec("abc")
    with pytest.raises(RuntimeError):
        cd.decode(np.zeros((4, 1, 5), np.float32))


def test_codec_surface_matches_reference(pkg):
    """Constructor state / encode of the drop-in codec (utils/ctc_codec.py:17-61)."""
    cd = pkg.ctc_codec("天地人")
    assert cd.characters == ["<blank>", "天", "地", "人", "<unknown>"]
    assert cd.dict["<blank>"] == 0 and cd.dict["<unknown>"] == 4 and cd.dict["地"] == 2
    assert (cd.lm_panelty, cd.len_bonus, cd.search_depth, cd.beam_size) == (2, 5.8, 10, 10)
    assert (cd.use_tfm_score, cd.use_tfm_pred, cd.skip_search, cd.use_beam_search) == (False, True, False, False)
    idx, ln = cd.encode(["天人?", "", "地"])
    assert idx.dtype == np.int32 and idx.tolist() == [1, 3, 4, 2] and ln.tolist() == [3, 0, 1]
    with open(os.path.join(GOLDEN, "codec_cases.json")) as f:
        gold = json.load(f)
    for name, seed, w, b, c, style in codec_cases.CODEC_CASES:
        chars = codec_cases.vocab(c)
        enc = pkg.ctc_codec(chars).encode(["".join(chars[:3]) + "?", "", chars[-1]])
        assert [enc[0].tolist(), enc[1].tolist()] == gold[name]["encode"]


def _frontend_numpy(logits, k):
    """What the device front end produces, computed with numpy for the host-search test only."""
    logp = log_softmax(logits, axis=2)
    W, B, C = logp.shape
    order = np.argsort(-logp, axis=2, kind="stable")[:, :, :k].astype(np.int32)
    fe = {"W": W, "B": B, "C": C, "k": k, "topk_idx": np.ascontiguousarray(order),
          "topk_logp": np.ascontiguousarray(np.take_along_axis(logp, order, axis=2)),
          "blank_logp": np.ascontiguousarray(logp[:, :, 0])}
    thresh = np.log(0.001)
    off, ci, cl = [0], [], []
    for t in range(W):
        for b in range(B):
            c = np.where(logp[t, b] > thresh)[0]
            ci.extend(c.tolist())
            cl.extend(logp[t, b, c].tolist())
            off.append(len(ci))
    fe["cand_off"] = np.array(off, dtype=np.int64)
    fe["cand_idx"] = np.array(ci + [0], dtype=np.int32)
    fe["cand_logp"] = np.array(cl + [0], dtype=np.float32)
    return fe


def test_host_beam_search_matches_reference(pkg):
    """csrc/beam_search.cpp (pure host code) vs the REAL reference codec's outputs: exact strings,
    for the built-in LMs (threaded) and for a Python LM object through the callback."""
    from oracle import ctc_ref
    with open(os.path.join(GOLDEN, "codec_cases.json")) as f:
        gold = json.load(f)
    for name, seed, w, b, c, style in codec_cases.CODEC_CASES:
        logits = codec_cases.gen_logits(seed, w, b, c, style)
        chars = codec_cases.vocab(c)
        for tag, skip, lm, lp, lb, bs, depth in codec_cases.BEAM_SETTINGS:
            for via_callback in (False, True):
                cd = pkg.ctc_codec(chars)
                cd.use_beam_search, cd.skip_search, cd.use_tfm_pred = True, skip, False
                cd.lm_panelty, cd.len_bonus, cd.beam_size, cd.search_depth = lp, lb, bs, depth
                if via_callback:
                    cd.ngram = ctc_ref.ZeroLM() if lm == "zero" else ctc_ref.ToyBigramLM()
                else:
                    cd.ngram = pkg.ZeroLM() if lm == "zero" else pkg.ToyBigramLM()
                try:
                    got = cd.decode_frontend(_frontend_numpy(logits, min(depth, c)))
                except IndexError:
                    got = "IndexError"
                assert got == gold[name][tag], (name, tag, via_callback)


def test_host_beam_search_lm_exception_propagates(pkg):
    logits = codec_cases.gen_logits(1, 20, 1, 12, "flat")

    class Boom(object):
        def score(self, sentence, eos=False):
            raise ZeroDivisionError("lm failed")

    cd = pkg.ctc_codec(codec_cases.vocab(12))
    cd.use_beam_search, cd.use_tfm_pred, cd.ngram = True, False, Boom()
    with pytest.raises(ZeroDivisionError):
        cd.decode_frontend(_frontend_numpy(logits, 10))


def test_host_beam_search_transformer_hooks(pkg):
    """use_tfm_score / use_tfm_pred duck-typed hooks (utils/ctc_codec.py:215-227,269-274): the engine's host search
    AND the oracle codec, each driven by the same fake transformer object, must reproduce the strings the REAL
    reference codec produced with it (tests/golden/codec_tfm.json, made by make_golden_tfm.py), including
    next_k_words lists shorter than k and LONGER than k (the reference chains whatever it gets, :225-226; the C ABI's
    callback then asks for more slots and is called again)."""
    from oracle import ctc_ref
    with open(os.path.join(GOLDEN, "codec_tfm.json")) as f:
        gold = json.load(f)
    distinct = set()
    for name, seed, w, b, c, style, score, pred, ragged in codec_cases.TFM_CASES:
        chars = codec_cases.vocab(c)
        logits = codec_cases.gen_logits(seed, w, b, c, style)
        depth = codec_cases.TFM_SETTINGS["search_depth"]
        oc = ctc_ref.CtcCodecRef(chars)
        cd = pkg.ctc_codec(chars)
        for obj in (oc, cd):
            obj.use_beam_search, obj.use_tfm_score, obj.use_tfm_pred = True, score, pred
            for k, v in codec_cases.TFM_SETTINGS.items():
                setattr(obj, k, v)
            obj.transformer, obj.ngram = codec_cases.FakeTransformer(list(chars), ragged), ctc_ref.ToyBigramLM()
        assert oc.decode(logits) == gold[name], ("oracle", name)
        full = np.ascontiguousarray(log_softmax(logits, axis=2), dtype=np.float32) if pred else None
        assert cd.decode_frontend(_frontend_numpy(logits, depth), full) == gold[name], ("engine", name)
        distinct.add(json.dumps(gold[name], ensure_ascii=False))
    assert len(distinct) >= 4          # the hooks change the result: the cases are not all the same decode


def test_shard_and_pack(pkg):
    from importlib import import_module
    dist = import_module(pkg.__name__ + ".dist")
    for n, world in ((4096, 8), (10, 4), (3, 8), (0, 2)):
        spans = [dist.shard_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    labs = [np.array([5, 6, 7], np.int32), np.array([], np.int32), np.array([1], np.int32)]
    back = dist.unpack_labels(dist.pack_labels(labs, 8))
    assert all(np.array_equal(a, b) for a, b in zip(labs, back))
    with pytest.raises(ValueError):
        dist.pack_labels([np.arange(9, dtype=np.int32)], 8)


GLOO_WORKER = r'''
import os, sys, importlib
import numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
pkg = importlib.import_module("handwritten-chinese-ocr-samples_amd")
d = importlib.import_module("handwritten-chinese-ocr-samples_amd.dist")
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n = 7
lines = [np.arange(1, 1 + (i * 3) %% 5, dtype=np.int32) + i for i in range(n)]
lo, hi = d.shard_range(n, rank, world)
out = d.gather_labels(lines[lo:hi], n, cap=8)
if rank == 0:
    assert len(out) == n and all(np.array_equal(a, b) for a, b in zip(out, lines)), out
    print("GATHER_OK")
else:
    assert out is None
dist.destroy_process_group()
'''


def test_gather_world_size_2_gloo(tmp_path):
    """N > 1 path on CPU: batch shards -> ONE gather -> global line order on rank 0."""
    script = tmp_path / "worker.py"
    script.write_text(GLOO_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29513", str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "GATHER_OK" in res.stdout


SHARD_WORKER = r'''
import os, sys, importlib
import numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
d = importlib.import_module("handwritten-chinese-ocr-samples_amd.dist")
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()

class FakeModel(object):            # stands in for hctr_model.greedy: labels depend on the line's pixels AND its pad width
    calls = []
    def greedy(self, images, widths=None):
        self.calls.append(images.shape)
        return [np.array([int(img[:, :w].sum()) %% 97 + 1, images.shape[2], int(w)], np.int32)
                for img, w in zip(images, widths)]

widths = np.array([30, 21, 13, 6, 2, 28, 9], np.int32)
rng = np.random.default_rng(3)
imgs = rng.integers(0, 255, (len(widths), 128, 30), dtype=np.uint8)
m = FakeModel()
got = d.recognize_sharded(m, imgs, widths)
lo, hi = d.shard_range(len(widths), rank, world)
assert m.calls == [(hi - lo, 128, 30)]             # one call, own shard, GLOBAL pad width
if rank == 0:
    want = FakeModel().greedy(imgs, widths)
    assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want)), got
    print("SHARD_OK")
else:
    assert got is None
dist.destroy_process_group()
'''


def test_recognize_sharded_world_size_2_gloo(tmp_path):
    """dist.recognize_sharded on CPU with a stand-in model: contiguous shards of the globally padded batch, one
    greedy call per rank, ONE gather, global line order on rank 0 (the engine itself: test_gpu_parity.py)."""
    script = tmp_path / "worker.py"
    script.write_text(SHARD_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29514", str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "SHARD_OK" in res.stdout


def test_plan_batches(pkg):
    from importlib import import_module
    bk = import_module(pkg.__name__ + ".bucketing")
    widths = [800] * 5 + [3200] * 3 + [1600] * 4 + [2400] * 2 + [3190]
    plan = bk.plan_batches(widths, max_lines=4, max_pad_fraction=0.1)
    flat = sorted(int(i) for b in plan for i in b)
    assert flat == list(range(len(widths)))                      # every line exactly once
    for b in plan:
        ws = [widths[i] for i in b]
        assert len(b) <= 4 and min(ws) >= 0.9 * max(ws)
    assert [widths[i] for i in plan[0]] == [3200, 3200, 3200, 3190]   # widest first, near-equal grouped
    imgs = [np.full((128, w), 7, np.uint8) for w in (5, 9, 9)]
    batch, ws = bk.pad_batch(imgs, [1, 0])
    assert batch.shape == (2, 128, 9) and ws.tolist() == [9, 5] and batch[1, 0, 5:].sum() == 0


def test_arpa_ngram_scorer(pkg, tmp_path):
    """csrc/ngram_lm.cpp vs the oracle's independent ARPA scorer and a hand-computed back-off case
    (kenlm.Model.score semantics: log10, <s> context, eos optional, OOV -> <unk>)."""
    from oracle import ctc_ref
    # hand-made model: P(b|a) present, P(c|a b) absent -> backs off via bow(a b) [absent: 0] to P(c|b)
    # [absent] -> bow(b) + P(c)
    arpa = tmp_path / "hand.arpa"
    arpa.write_text("\\data\\\nngram 1=6\nngram 2=2\nngram 3=1\n\n\\1-grams:\n"
                    "-2.0\t<unk>\n-1.5\t<s>\t-0.5\n-1.2\t</s>\n-0.7\ta\t-0.3\n-0.9\tb\t-0.25\n-1.1\tc\t-0.1\n\n"
                    "\\2-grams:\n-0.4\t<s> a\t-0.2\n-0.6\ta b\t-0.15\n\n\\3-grams:\n-0.05\t<s> a b\n\n\\end\\\n",
                    encoding="utf-8")
    lm = pkg.ArpaLM(str(arpa))
    assert lm.order == 3
    # <s> a b c : P(a|<s>) = -0.4 ; P(b|<s> a) = -0.05 (trigram) ; P(c|a b): no "a b c", bow(a b) = -0.15,
    # no "b c", bow(b) = -0.25, P(c) = -1.1  => -1.5
    assert abs(lm.score("a b c", eos=False) - (-0.4 - 0.05 - 0.15 - 0.25 - 1.1)) < 1e-6
    assert abs(lm.score("a", bos=False, eos=False) - (-0.7)) < 1e-6
    assert abs(lm.score("zzz", bos=False, eos=False) - (-2.0)) < 1e-6             # OOV -> <unk>
    # </s> after "<s> a": no "<s> a </s>" -> bow(<s> a) = -0.2; no "a </s>" -> bow(a) = -0.3; P(</s>) = -1.2
    assert abs(lm.score("a", eos=True) - (-0.4 - 0.2 - 0.3 - 1.2)) < 1e-6
    ref = ctc_ref.ArpaRef(str(arpa))
    for sent in ("a b c", "c c a b", "", "b zzz a", "a a a a a"):
        for bos in (True, False):
            for eos in (True, False):
                assert abs(lm.score(sent, bos=bos, eos=eos) - ref.score(sent, bos=bos, eos=eos)) < 1e-9, (sent, bos, eos)
    toy = codec_cases.write_toy_arpa(str(tmp_path / "toy.arpa"))
    lm, ref = pkg.ArpaLM(toy), ctc_ref.ArpaRef(toy)
    chars = codec_cases.vocab(18)
    rng = np.random.RandomState(3)
    for _ in range(200):
        sent = " ".join(chars[i] for i in rng.randint(0, 16, size=rng.randint(0, 12)))
        assert abs(lm.score(sent, eos=False) - ref.score(sent, eos=False)) < 1e-9
    with pytest.raises(OSError):
        pkg.ArpaLM(str(tmp_path / "missing.arpa"))


def test_beam_search_with_native_arpa_lm(pkg, tmp_path):
    """Beam search with the built-in ARPA model (threaded, no callbacks) == the oracle codec driven by
    the oracle ARPA scorer == the same model through the Python callback path. Exact strings."""
    from oracle import ctc_ref
    toy = codec_cases.write_toy_arpa(str(tmp_path / "toy.arpa"))
    c = 16
    chars = codec_cases.vocab(c)
    for seed, w, style, skip in ((21, 60, "mixed", False), (22, 45, "flat", False), (23, 60, "mixed", True)):
        logits = codec_cases.gen_logits(seed, w, 2, c, style)
        oc = ctc_ref.CtcCodecRef(chars)
        oc.use_beam_search, oc.use_tfm_pred, oc.skip_search = True, False, skip
        oc.lm_panelty, oc.len_bonus, oc.ngram = 0.8, 4.8, ctc_ref.ArpaRef(toy)
        want = oc.decode(logits)
        fe = _frontend_numpy(logits, 10)
        for lm in (pkg.ArpaLM(toy), ctc_ref.ArpaRef(toy)):      # native builtin path, callback path
            cd = pkg.ctc_codec(chars)
            cd.use_beam_search, cd.use_tfm_pred, cd.skip_search = True, False, skip
            cd.lm_panelty, cd.len_bonus, cd.ngram = 0.8, 4.8, lm
            assert cd.decode_frontend(fe) == want, (seed, type(lm).__name__)
    cd = pkg.ctc_codec(chars)
    cd.set_beam_search(ngram_path=toy, use_tfm_pred=False, lm_panelty=0.8, len_bonus=4.8)   # -kp file.arpa
    assert isinstance(cd.ngram, pkg.ArpaLM)


def test_host_beam_search_fuzz(pkg):
    """Property test (hypothesis): for random shapes, logits styles and beam settings the C++ search
    returns exactly the oracle codec's strings (or raises IndexError exactly when the oracle does)."""
    from hypothesis import given, settings, strategies as st
    from oracle import ctc_ref

    @settings(max_examples=40, deadline=None)
    @given(seed=st.integers(0, 10 ** 6), w=st.integers(1, 40), b=st.integers(1, 3), c=st.integers(3, 40),
           style=st.sampled_from(["peaky", "flat", "mixed"]), skip=st.booleans(),
           lm=st.sampled_from(["zero", "toy"]), beam=st.integers(1, 12), depth=st.integers(1, 12),
           lp=st.floats(0.0, 3.0), lb=st.floats(0.0, 8.0))
    def run(seed, w, b, c, style, skip, lm, beam, depth, lp, lb):
        logits = codec_cases.gen_logits(seed, w, b, c, style)
        chars = codec_cases.vocab(c)
        oc = ctc_ref.CtcCodecRef(chars)
        oc.use_beam_search, oc.use_tfm_pred, oc.skip_search = True, False, skip
        oc.beam_size, oc.search_depth, oc.lm_panelty, oc.len_bonus = beam, depth, lp, lb
        oc.ngram = ctc_ref.ZeroLM() if lm == "zero" else ctc_ref.ToyBigramLM()
        try:
            want = oc.decode(logits)
        except IndexError:
            want = "IndexError"
        cd = pkg.ctc_codec(chars)
        cd.use_beam_search, cd.use_tfm_pred, cd.skip_search = True, False, skip
        cd.beam_size, cd.search_depth, cd.lm_panelty, cd.len_bonus = beam, depth, lp, lb
        cd.ngram = pkg.ZeroLM() if lm == "zero" else pkg.ToyBigramLM()
        try:
            got = cd.decode_frontend(_frontend_numpy(logits, min(depth, c)))
        except IndexError:
            got = "IndexError"
        assert got == want

    run()


def test_header_and_c_example_compile_as_c99(tmp_path):
    """include/hctr_hip.h is a C header (extern "C", plain pointers): it and examples/greedy_demo.c compile with
    gcc -std=c99 -pedantic, and the example links against the in-tree library."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    src = tmp_path / "t.c"
    src.write_text('#include "hctr_hip.h"\nint main(void) { hctr_beam_params p; (void)p; return hctr_version() == 0; }\n')
    inc = os.path.join(ROOT, "include")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-c", str(src), "-o",
                        str(tmp_path / "t.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    from conftest import ROOT as root
    libdir = os.path.join(root, "handwritten-chinese-ocr-samples_amd")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc,
                        os.path.join(root, "examples", "greedy_demo.c"), "-L", libdir, "-lhctr_hip",
                        "-Wl,-rpath," + libdir, "-o", str(tmp_path / "greedy_demo")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(tmp_path / "greedy_demo")], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr          # runs (no GPU needed to print usage)


def test_abi_no_exception_crosses():
    """include/hctr_hip.h: "no C++ exception crosses the ABI". A child process lowers RLIMIT_AS so that heap growth
    (std::bad_alloc in the prefix trie) and thread creation (std::system_error) fail inside hctr_beam_search: the call
    must come back with HCTR_ERR_NOMEM instead of ending the process through std::terminate; with room it succeeds."""
    child = os.path.join(ROOT, "tests", "abi_guard_child.py")

    def run(headroom_mb, threads):
        r = subprocess.run([sys.executable, child, str(headroom_mb), str(threads)], capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0, (headroom_mb, threads, r.returncode, r.stderr[-400:])
        m = re.search(r"RC (-?\d+) LEN (\d+)", r.stdout)
        assert m, r.stdout
        return int(m.group(1)), int(m.group(2))

    assert run(4, 1) == (-7, 0)            # HCTR_ERR_NOMEM from the search's own allocations
    assert run(4, 8)[0] == -7              # no room for a thread stack either: falls back to the caller's thread
    rc, total = run(4096, 8)
    assert rc == 0 and total > 0


def test_state_dict_round_trip_without_gpu(pkg, synth):
    """``model.state_dict()`` hands back the loaded checkpoint in the reference's key schema (254 entries,
    main.py:349-356), so a caller can save / reload it like the reference's nn.Module (no GPU needed: the weights wait
    for ``.cuda()``)."""
    import torch
    C = 40
    sd = synth.make_state_dict(C, seed=3, calib=None)
    m = pkg.hctr_model(C)
    with pytest.raises(RuntimeError):
        m.state_dict()
    m.load_state_dict(sd)
    back = m.state_dict()
    assert list(back.keys()) == list(sd.keys()) and len(back) == 254
    assert all(isinstance(v, torch.Tensor) and np.array_equal(v.numpy(), sd[k]) for k, v in back.items())
    m2 = pkg.hctr_model(C)
    m2.load_state_dict(back)                               # torch tensors are accepted like numpy arrays
    assert np.array_equal(m2.state_dict()["linear.weight"].numpy(), sd["linear.weight"])


def test_bench_multi_gpu_launch_fails_loudly_without_gpus():
    """`python bench.py --gpus 2` starts its own ranks as child processes (torch.distributed.run) and exits with THEIR
    code: on a machine without a GPU the ranks fail at hctr_create (no CPU fallback), the parent prints no result line
    and returns non-zero - it neither hangs nor reports a number."""
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HCTR_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--batch", "2", "--width", "64"], env=env, capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by tests/test_gpu_parity.py::test_bench_strong_scaling_rehearsal")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "starting 2 ranks" in r.stderr


def test_beam_pipeline_chunk_schedule(pkg):
    """pipeline.chunk_schedule: contiguous cover of the batch, chunks never above the limit, large chunks first and a
    taper to small ones at the end (nothing overlaps the host search of the last chunk)."""
    from importlib import import_module
    pipe = import_module(pkg.__name__ + ".pipeline")
    for n, c in ((256, 64), (256, 32), (10, 64), (64, 64), (100, 32), (7, 2), (300, 128), (0, 64), (1, 64), (40, 64)):
        sp = pipe.chunk_schedule(n, c)
        assert sum(h - l for l, h in sp) == n and all(0 < h - l <= c for l, h in sp)
        assert all(a[1] == b[0] for a, b in zip(sp, sp[1:])) and (not sp or (sp[0][0] == 0 and sp[-1][1] == n))
    assert [h - l for l, h in pipe.chunk_schedule(256, 64)] == [64, 64, 64, 32, 16, 16]


def test_hot_kernels_do_not_spill(pkg, tmp_path):
    """The K loops of the hot kernels must not spill: a spill reload in front of an LDS-DMA drains vmcnt and serialises
    the loop. The 3x3 kernel sits at the 256-register limit, where an innocent edit flips the
    allocator - so the built library's own metadata is checked: no spilled vector register and no private segment for
    the instances the f16 forward pass launches (3x3 halo kernel both tile geometries + fused downsample, fused stem,
    head GEMM)."""
    import shutil
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(llvm + "/llvm-objdump") and os.path.exists(llvm + "/llvm-readelf")):
        pytest.skip("llvm-objdump / llvm-readelf not found")
    lib = tmp_path / "lib.so"
    shutil.copy(pkg.build(), lib)
    subprocess.run([llvm + "/llvm-objdump", "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    meta = {}
    for co in tmp_path.glob("lib.so.*gfx950"):
        notes = subprocess.run([llvm + "/llvm-readelf", "--notes", str(co)], check=True, capture_output=True, text=True).stdout
        cur = None
        for line in notes.splitlines():
            m = re.match(r"\s*\.(name|private_segment_fixed_size|vgpr_spill_count|vgpr_count):\s+(\S+)", line)
            if not m:
                continue
            if m.group(1) == "name":
                cur = meta.setdefault(m.group(2), {})
            elif cur is not None:
                cur[m.group(1)] = int(m.group(2))
    hot = ["conv3x3_halo4_kernelILi0ELb0ELb0ELb0ELb0E", "conv3x3_halo4_kernelILi1ELb0ELb0ELb0ELb0E",
           "conv3x3_halo4_kernelILi0ELb0ELb0ELb0ELb1E", "stem_conv0_2_kernel", "conv_mfma_kernelILi2ELi4ELi8ELi1ELb1E"]
    for h in hot:
        names = [n for n in meta if h in n]
        assert names, h
        for n in names:
            assert meta[n]["vgpr_spill_count"] == 0 and meta[n]["private_segment_fixed_size"] == 0, (n, meta[n])
