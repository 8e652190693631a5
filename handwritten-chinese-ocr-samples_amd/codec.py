"""Drop-in for the reference's ``ctc_codec`` (utils/ctc_codec.py:14-307) on the gfx950 engine.

Same constructor, public attributes (including the reference's ``lm_panelty`` spelling),
``encode`` / ``decode`` / ``set_beam_search`` signatures. ``decode`` takes the ``[W, B, C]`` logits
(numpy, or a torch tensor that may already live on the GPU) and runs:

  greedy       argmax + CTC collapse kernels                       (utils/ctc_codec.py:70-99)
  beam search  device log-softmax / top-k / candidate lists, then the C++ host prefix search
               (csrc/beam_search.cpp) with the language model behind callbacks (:124-285)

Language models stay duck-typed exactly as in the reference: ``codec.ngram`` needs
``.score(sentence, eos=False)``, ``codec.transformer`` needs ``.score(list, char_based=True)`` and
``.next_k_words(list, k=, char_based=True)``. Nothing here runs a CPU re-implementation of the
decode: without a GPU the calls raise.
"""
import ctypes
import os

import numpy as np

from . import _lib
from .model import beam_frontend_call


class ZeroLM(object):
    """Language model that scores every sentence 0 (built into the C++ search)."""

    def score(self, sentence, eos=False):
        return 0.0


class ToyBigramLM(object):
    """Deterministic hashed character-bigram LM used by tests and benchmarks (built into the C++
    search, formula shared with oracle/ctc_ref.py): exercises the LM term without kenlm."""

    def score(self, sentence, eos=False):
        s, prev = 0.0, 0
        for ch in sentence.split(" "):
            if ch == "":
                continue
            c = ord(ch)
            h = (prev * 2654435761 + c * 40503 + 12345) & 0xFFFFFFFF
            h ^= h >> 15
            h = (h * 2246822519) & 0xFFFFFFFF
            h ^= h >> 13
            s += -4.0 * ((h & 0xFFFF) / 65536.0)
            prev = c
        return s


class ArpaLM(object):
    """ARPA back-off n-gram model scored natively (csrc/ngram_lm.cpp) with the interface the reference
    expects from ``kenlm.Model``: ``score(sentence, bos=True, eos=True)`` = log10 probability of a
    space-separated sentence. Beam search uses it without Python callbacks (multi-threaded)."""

    def __init__(self, arpa_path):
        lib = _lib.load()
        h = ctypes.c_void_p()
        rc = lib.hctr_ngram_load(str(arpa_path).encode("utf-8"), ctypes.byref(h))
        if rc != 0:
            raise OSError("cannot load ARPA model: %s" % lib.hctr_ngram_last_error().decode("utf-8", "replace"))
        self._h = h
        self.path = str(arpa_path)
        self.order = lib.hctr_ngram_order(h)

    def score(self, sentence, bos=True, eos=True):
        return float(_lib.load().hctr_ngram_score(self._h, sentence.encode("utf-8"), int(bos), int(eos)))

    def word_id(self, token):
        return int(_lib.load().hctr_ngram_word_id(self._h, token.encode("utf-8")))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().hctr_ngram_free(self._h)
                self._h = None
        except Exception:
            pass


class ctc_codec(object):
    """ Convert between text-label and text-index """

    def __init__(self, characters_str):
        # utils/ctc_codec.py:17-41
        self.chars_list = list(characters_str)
        self.dict = {}
        for i, char in enumerate(self.chars_list):
            self.dict[char] = i + 1                      # 0 is the CTC blank
        self.characters = ['<blank>'] + self.chars_list + ['<unknown>']
        self.dict['<blank>'] = 0
        self.dict['<unknown>'] = len(self.characters) - 1

        self.ngram = None
        self.transformer = None
        self.lm_panelty = 2
        self.len_bonus = 5.8
        self.search_depth = 10
        self.beam_size = 10
        self.use_tfm_score = False
        self.use_tfm_pred = True
        self.skip_search = False
        self.use_beam_search = False

        self.num_threads = 0          # host beam-search threads (0 = automatic, see decode_frontend); built-in LMs only
        self._ctx = None
        self._own_ctx = False
        self._model = None            # attach(): resolve the engine context through the model at call time
        self._device = 0

    # -- engine binding -----------------------------------------------------------------------
    def cuda(self, device=0):
        """Choose the GPU the decode kernels run on (default 0)."""
        self._drop_ctx()
        self._device = int(device)
        return self

    def attach(self, model):
        """Share an hctr_model's engine context (saves a second context on the same GPU). The context is
        looked up through the model on every call, so moving or releasing the model never leaves a
        dangling pointer here."""
        self._drop_ctx()
        if model._ctx is None:
            raise RuntimeError("model is not on a GPU")
        self._model = model
        return self

    def _context(self):
        if self._model is not None:
            if self._model._ctx is None:
                raise RuntimeError("the attached hctr_model is no longer on a GPU")
            return self._model._ctx
        if self._ctx is None:
            ctx = ctypes.c_void_p()
            _lib.check(_lib.load().hctr_create(ctypes.byref(ctx), self._device, max(3, len(self.characters))))
            self._ctx, self._own_ctx = ctx, True
        return self._ctx

    def _drop_ctx(self):
        if self._ctx is not None and self._own_ctx:
            _lib.load().hctr_destroy(self._ctx)
        self._ctx, self._own_ctx, self._model = None, False, None

    def __del__(self):
        try:
            self._drop_ctx()
        except Exception:
            pass

    # -- encode: utils/ctc_codec.py:43-61 ---------------------------------------------------------
    def encode(self, text):
        length = [len(s) for s in text]
        unknown = len(self.characters) - 1
        known = self.dict
        index = [known[ch] if (ch in known and len(ch) == 1) else unknown for s in text for ch in s]
        return (np.array(index, dtype=np.int32), np.array(length, dtype=np.int32))

    # -- decode: utils/ctc_codec.py:63-68 ---------------------------------------------------------
    def decode(self, preds):
        logits, on_dev = self._as_logits(preds)
        W, B, C = (int(v) for v in logits.shape)
        if C != len(self.characters):
            raise ValueError("logits have %d classes, codec has %d" % (C, len(self.characters)))
        if not self.use_beam_search:
            if W == 0:
                return []                                # reference skips zero-length lines (:85-86)
            ctx = self._context()
            labels = np.empty((B, W), dtype=np.int32)
            lengths = np.empty((B,), dtype=np.int32)
            _lib.check(_lib.load().hctr_decode_greedy_logits(ctx, _lib.ptr(logits), on_dev, W, B, C,
                                                             _lib.ptr(labels), _lib.ptr(lengths)), ctx)
            return self.labels_to_text([labels[b, :lengths[b]] for b in range(B)])
        ctx = self._context()
        k = min(int(self.search_depth), C)
        fe = beam_frontend_call(ctx, None, _lib.F32, 0, None, logits, on_dev, B, W, C, k, bool(self.skip_search))
        full_logp = None
        if self.use_tfm_pred:
            full_logp = self._full_logp(logits, on_dev)
        return self.decode_frontend(fe, full_logp)

    def labels_to_text(self, label_lists):
        chars = self.characters
        return ["".join(chars[i] for i in line) for line in label_lists]

    @staticmethod
    def _as_logits(preds):
        if hasattr(preds, "data_ptr"):                  # torch tensor (possibly on the GPU)
            import torch
            t = preds.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.float().contiguous()
            if t.is_cuda:
                torch.cuda.current_stream(t.device).synchronize()
            if t.dim() != 3:
                raise ValueError("preds must be [W,B,C]")
            return t, int(t.is_cuda)
        a = np.ascontiguousarray(preds, dtype=np.float32)
        if a.ndim != 3:
            raise ValueError("preds must be [W,B,C]")
        return a, 0

    def _full_logp(self, logits, on_dev):
        # only for LM-proposed candidates (use_tfm_pred): the search may then look up any class.
        W, B, C = (int(v) for v in logits.shape)
        out = np.empty((W, B, C), dtype=np.float32)
        ctx = self._context()
        _lib.check(_lib.load().hctr_log_softmax(ctx, _lib.ptr(logits), on_dev, W, B, C, _lib.ptr(out)), ctx)
        return out

    # -- beam search over a device front end ------------------------------------------------------
    def decode_frontend(self, fe, full_logp=None):
        """Run the host prefix beam search on ``hctr_model.beam_frontend`` /
        ``hctr_beam_frontend`` output and return the decoded strings."""
        W, B, C, k = fe["W"], fe["B"], fe["C"], fe["k"]
        if B == 0:
            return []
        lib = _lib.load()
        params = _lib.BeamParams()
        params.skip_search = int(bool(self.skip_search))
        params.beam_size = int(self.beam_size)
        params.search_depth = min(int(self.search_depth), k)
        params.lm_panelty = float(self.lm_panelty)
        params.len_bonus = float(self.len_bonus)
        # default: the CPUs of the affinity mask, at most 64 and at most 4x the cgroup CPU quota (measured on the GPU box,
        # quota 16 of 256 logical CPUs, 64 lines per call: 16 threads 73 ms, 32 threads 45 ms, 64 threads 44 ms, 128
        # threads 100 ms - short bursts run ahead of the quota, more threads than that only contend)
        params.num_threads = int(self.num_threads) or min(64, len(os.sched_getaffinity(0)), 4 * _lib.usable_cpus())
        params.user = None
        chars = self.characters
        err = []
        keep = []                                        # keep callbacks / arrays alive during the call

        use_tfm_score = bool(self.use_tfm_score)
        use_tfm_pred = bool(self.use_tfm_pred)
        lm = self.transformer if use_tfm_score else self.ngram
        if not use_tfm_score and isinstance(lm, ZeroLM):
            params.builtin_lm = 1
        elif not use_tfm_score and isinstance(lm, ArpaLM):
            params.builtin_lm = 3
            unk = lm.word_id("<unk>")                    # OOV labels take <unk>'s id (as kenlm's vocabulary does)
            words = np.array([(lm.word_id(c) if lm.word_id(c) >= 0 else unk) if len(c) == 1 else unk
                              for c in chars], dtype=np.int32)
            keep.append(words)
            params.ngram = lm._h
            params.label_words = words.ctypes.data
        elif not use_tfm_score and isinstance(lm, ToyBigramLM):
            params.builtin_lm = 2
            cps = np.array([ord(c) if len(c) == 1 else 0 for c in chars], dtype=np.int32)
            keep.append(cps)
            params.label_codepoints = cps.ctypes.data
        else:
            if lm is None:
                raise RuntimeError("beam search needs a language model: set codec.ngram (or "
                                   "codec.transformer with use_tfm_score) - utils/ctc_codec.py:267-281")
            params.builtin_lm = 0

            def score_cb(user, n, ids, offs, scores):
                try:
                    sents = [[chars[ids[j]] for j in range(offs[i], offs[i + 1])] for i in range(n)]
                    if use_tfm_score:
                        vals = lm.score(["".join(s) for s in sents], char_based=True)
                        for i in range(n):
                            scores[i] = float(vals[i])
                    else:
                        for i in range(n):
                            scores[i] = float(lm.score(" ".join(sents[i]), eos=False))
                    return 0
                except BaseException as exc:             # never unwind through C
                    err.append(exc)
                    return -100

            cb = _lib.LM_SCORE_CB(score_cb)
            keep.append(cb)
            params.score_cb = cb

        if use_tfm_pred:
            tfm = self.transformer
            if tfm is None:
                raise RuntimeError("use_tfm_pred=True needs codec.transformer (utils/ctc_codec.py:215-219)")
            if full_logp is None:
                raise ValueError("use_tfm_pred needs the full log-prob tensor")
            depth = params.search_depth
            cdict = self.dict

            last = {}                                    # the LM's answer of a call that asked for more slots

            def next_cb(user, n, ids, offs, kk, out_ids):
                try:
                    key = (n, tuple(ids[j] for j in range(offs[n])), tuple(offs[i] for i in range(n + 1)))
                    if last.get("key") == key:
                        words = last["words"]
                    else:
                        prefixes = ["".join(chars[ids[j]] for j in range(offs[i], offs[i + 1])) for i in range(n)]
                        words = [list(w) for w in tfm.next_k_words(prefixes, k=depth, char_based=True)]
                    # the reference chains whatever the LM returns (utils/ctc_codec.py:225-226): a shorter list is
                    # padded with the <unknown> id, which the search skips like the reference does (:238-239); a
                    # longer one makes the search call again with as many slots as the longest list needs
                    need = max([len(w) for w in words] + [0])
                    if need > kk:
                        last["key"], last["words"] = key, words
                        return need
                    last.clear()
                    unk = len(chars) - 1
                    for i in range(n):
                        wl = words[i]
                        for j in range(kk):
                            out_ids[i * kk + j] = cdict[wl[j]] if j < len(wl) else unk
                    return 0
                except BaseException as exc:
                    err.append(exc)
                    return -100

            ncb = _lib.LM_NEXT_CB(next_cb)
            keep.append(ncb)
            params.next_cb = ncb

        labels = np.zeros((B, max(W, 1)), dtype=np.int32)
        lengths = np.zeros((B,), dtype=np.int32)
        status = np.zeros((B,), dtype=np.int32)
        rc = lib.hctr_beam_search(ctypes.byref(params), W, B, C, k, _lib.ptr(fe["topk_idx"]),
                                  _lib.ptr(fe["topk_logp"]), _lib.ptr(fe["blank_logp"]),
                                  _lib.ptr(fe["cand_off"]), _lib.ptr(fe["cand_idx"]), _lib.ptr(fe["cand_logp"]),
                                  _lib.ptr(full_logp), _lib.ptr(labels), _lib.ptr(lengths), _lib.ptr(status))
        if err:
            raise err[0]
        if rc == _lib.ERR_EMPTY_LINE:
            raise IndexError("list index out of range (beam search on a line with an empty greedy decode "
                             "or an emptied beam set; reference utils/ctc_codec.py:143,179,198,208)")
        if rc != 0:
            raise RuntimeError("hctr_beam_search failed with status %d" % rc)
        return self.labels_to_text([labels[b, :lengths[b]] for b in range(B)])

    # -- set_beam_search: utils/ctc_codec.py:101-122 ------------------------------------------------
    def set_beam_search(self, skip_search=False, ngram_path='', tfm_path='',
                        lm_panelty=2, len_bonus=5.8, beam_size=10, search_depth=10,
                        use_tfm_score=False, use_tfm_pred=True,
                        use_openvino=False):
        self.use_beam_search = True
        self.lm_panelty = lm_panelty
        self.len_bonus = len_bonus
        self.beam_size = beam_size
        self.search_depth = search_depth
        self.use_tfm_pred = use_tfm_pred
        self.use_tfm_score = use_tfm_score
        self.skip_search = skip_search
        if use_tfm_pred or use_tfm_score:
            # The reference loads a fairseq / OpenVINO transformer here (utils/transformer_infer.py);
            # neither is part of this engine. Attach any object with the same
            # duck-typed interface as ``codec.transformer`` instead.
            if self.transformer is None:
                raise ImportError("transformer language models are not bundled: assign codec.transformer "
                                  "(needs .score(list, char_based=True) / .next_k_words(list, k=, char_based=True)) "
                                  "before set_beam_search, or pass use_tfm_pred=False, use_tfm_score=False")
        if not use_tfm_score:
            if ngram_path in ('zero', 'builtin:zero'):
                self.ngram = ZeroLM()
            elif ngram_path in ('toy', 'builtin:toy'):
                self.ngram = ToyBigramLM()
            elif str(ngram_path).lower().endswith(".arpa"):
                self.ngram = ArpaLM(ngram_path)           # native ARPA scorer, no kenlm needed
            elif self.ngram is None or ngram_path:
                import kenlm                              # same optional dependency as the reference
                self.ngram = kenlm.Model(ngram_path)
