"""Drop-in for the reference's ``hctr_model`` (models/handwritten_ctr_model.py:156-178),
backed by the gfx950 engine through the C ABI (include/hctr_hip.h).

Same constructor, attributes and call signature; ``.cuda(idx)`` binds the engine to a GPU,
``load_state_dict`` ingests a reference checkpoint dict (test.py:152-153), ``model(x)`` returns the
``[W, B, C]`` float32 logits. There is no CPU execution path: calling a model that was never moved
to a GPU raises, it does not fall back.

Beyond the reference surface the class exposes the fused fast paths the engine is built for
(``greedy`` / ``beam_frontend``), which keep the 29 kB-per-column logits on the device.
"""
import collections
import ctypes

import numpy as np

from . import _lib

IncompatibleKeys = collections.namedtuple("IncompatibleKeys", ["missing_keys", "unexpected_keys"])
_PRECISIONS = {"f16": 0, "f16x3": 1, "auto": 2}


def _is_torch(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


class hctr_model(object):
    def __init__(self, num_classes=7375, precision="f16"):
        """``precision`` (an engine option, not part of the reference surface): "f16" (default; fp16 storage/MFMA, fp32
        accumulate), "f16x3" (hi+lo split pairs, ~3x the work, fp32-grade logits) or "auto" (guarded: every line in f16,
        and the lines with a column whose top-1/top-2 logit margin is within twice the f16 logit tolerance once more in
        f16x3 - the f16x3 mode's text wherever f16 cannot certify its own, f16 speed on peaky logits; see
        include/hctr_hip.h ``hctr_set_precision``). A model built with "auto" holds both weight sets and can be switched
        between all three modes afterwards (``set_precision``)."""
        if precision not in _PRECISIONS:
            raise ValueError("precision must be 'f16', 'f16x3' or 'auto'")
        self.precision = precision
        # attributes of the reference class (models/handwritten_ctr_model.py:159-164)
        self.img_height = 128
        self.PAD = 'NormalizePAD'
        self.optimizer = 'SGD'
        self.pred = 'CTC'
        self.noutput = num_classes
        self.training = False
        self._ctx = None
        self._device = None
        self._pending_sd = None
        self._sd_host = None           # host copy of the loaded checkpoint (device moves, like nn.Module.cuda)
        self._loaded = False

    # -- nn.Module surface used by test.py ------------------------------------------------
    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("the hctr engine is inference-only (training is out of scope)")
        return self.eval()

    def cpu(self):
        # test.py:146-148. The engine is GPU-only; a later forward raises instead of falling back.
        if self._loaded:
            self._pending_sd = self._sd_host            # a later .cuda() brings the weights back
        self._release()
        self._device = None
        return self

    def cuda(self, device=None):
        if device is None:
            device = 0
        if hasattr(device, "index"):
            device = device.index or 0
        device = int(device)
        if self._ctx is not None and self._device == device:
            return self
        if self._loaded and self._pending_sd is None:
            self._pending_sd = self._sd_host            # moving a loaded model: re-ingest on the new device
        self._release()
        lib = _lib.load()
        ctx = ctypes.c_void_p()
        _lib.check(lib.hctr_create(ctypes.byref(ctx), device, int(self.noutput)))
        _lib.check(lib.hctr_set_precision(ctx, _PRECISIONS[self.precision]), ctx)
        self._ctx, self._device = ctx, device
        if self._pending_sd is not None:
            sd, self._pending_sd = self._pending_sd, None
            self._ingest(sd)
        return self

    def to(self, device):
        s = str(device)
        if s.startswith("cuda"):
            return self.cuda(int(s.split(":")[1]) if ":" in s else 0)
        return self.cpu()

    def load_state_dict(self, state_dict, strict=True):
        """Reference checkpoint ingest (``checkpoint['state_dict']``, test.py:152-153)."""
        if not strict:
            raise NotImplementedError("only strict=True is supported")
        sd = {}
        for k, v in state_dict.items():
            if _is_torch(v):
                v = v.detach().cpu().numpy()
            sd[k] = np.ascontiguousarray(v).reshape(np.shape(v))     # (ascontiguousarray makes 0-d entries 1-d)
        if self._ctx is None:
            self._pending_sd = sd          # uploaded when .cuda() binds a device
        else:
            if self._loaded:               # the C context finalises once: rebuild it
                dev = self._device
                self._release()
                self.cuda(dev)
            self._ingest(sd)
        return IncompatibleKeys([], [])

    def _ingest(self, sd):
        lib = _lib.load()
        for k, a in sd.items():
            if a.dtype == np.int64:
                dt = _lib.I64
            elif a.dtype == np.float32:
                dt = _lib.F32
            else:
                a = a.astype(np.float32)
                dt = _lib.F32
            shape = (ctypes.c_int64 * max(1, a.ndim))(*a.shape)
            _lib.check(lib.hctr_load_tensor(self._ctx, k.encode("utf-8"), _lib.ptr(a), shape, a.ndim, dt), self._ctx)
        _lib.check(lib.hctr_finalize_weights(self._ctx), self._ctx)
        self._loaded = True
        self._sd_host = sd
        active = getattr(self, "_active_precision", None)
        if active is not None and active != self.precision:       # a mode switched at run time survives a device move
            _lib.check(lib.hctr_set_precision(self._ctx, _PRECISIONS[active]), self._ctx)

    def state_dict(self):
        """The checkpoint dict this model was loaded from (reference key schema, ``main.py:349-356``), as torch CPU
        tensors when torch is importable, else numpy arrays. The device copy lives in kernel layouts (BatchNorm folded,
        fp16 MFMA row order) and is not read back."""
        if self._sd_host is None and self._pending_sd is None:
            raise RuntimeError("hctr_model has no weights: call load_state_dict(...) first")
        sd = self._sd_host if self._sd_host is not None else self._pending_sd
        try:
            import torch
            return collections.OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in sd.items())
        except ImportError:
            return collections.OrderedDict((k, np.array(v)) for k, v in sd.items())

    def _release(self):
        if self._ctx is not None:
            _lib.load().hctr_destroy(self._ctx)
            self._ctx = None
            self._loaded = False

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # -- helpers -----------------------------------------------------------------------------
    def _require_ctx(self):
        if self._ctx is None:
            raise RuntimeError("hctr_model is not on a GPU: call .cuda(idx) first "
                               "(the MI355X engine has no CPU execution path)")
        if not self._loaded:
            raise RuntimeError("hctr_model has no weights: call load_state_dict(...) first")
        return self._ctx

    @staticmethod
    def _img_args(x):
        """(array-like, dtype code, on_device, B, W) for float [B,1,128,W] / uint8 [B,128,W]."""
        if _is_torch(x):
            import torch
            if x.dtype == torch.uint8:
                if x.dim() != 3 or x.shape[1] != 128:
                    raise ValueError("uint8 input must be [B,128,W]")
                x = x.contiguous()
                if x.is_cuda:        # the engine runs on its own stream: torch work that produced x must be done
                    torch.cuda.current_stream(x.device).synchronize()
                return x, _lib.U8, int(x.is_cuda), x.shape[0], x.shape[2]
            if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != 128:
                raise ValueError("input must be float [B,1,128,W], got %s" % (tuple(x.shape),))
            x = x.float().contiguous()
            if x.is_cuda:
                torch.cuda.current_stream(x.device).synchronize()
            return x, _lib.F32, int(x.is_cuda), x.shape[0], x.shape[3]
        x = np.asarray(x)
        if x.dtype == np.uint8:
            if x.ndim != 3 or x.shape[1] != 128:
                raise ValueError("uint8 input must be [B,128,W]")
            return np.ascontiguousarray(x), _lib.U8, 0, x.shape[0], x.shape[2]
        if x.ndim != 4 or x.shape[1] != 1 or x.shape[2] != 128:
            raise ValueError("input must be float [B,1,128,W], got %s" % (x.shape,))
        return np.ascontiguousarray(x, dtype=np.float32), _lib.F32, 0, x.shape[0], x.shape[3]

    @staticmethod
    def _widths(widths, B):
        if widths is None:
            return None
        w = np.ascontiguousarray(widths, dtype=np.int32)
        if w.shape != (B,):
            raise ValueError("widths must have shape [B]")
        return w

    # -- forward: models/handwritten_ctr_model.py:171-178 -----------------------------------------
    def forward(self, input, widths=None):
        ctx = self._require_ctx()
        x, dt, on_dev, B, W = self._img_args(input)
        wd = self._widths(widths, B)
        C = int(self.noutput)
        lib = _lib.load()
        if _is_torch(input):
            import torch
            out = torch.empty((W, B, C), dtype=torch.float32, device=input.device)
            out_dev = int(out.is_cuda)
        else:
            out = np.empty((W, B, C), dtype=np.float32)
            out_dev = 0
        _lib.check(lib.hctr_forward_logits(ctx, _lib.ptr(x), dt, on_dev, _lib.ptr(wd), B, W, _lib.ptr(out), out_dev), ctx)
        return out

    __call__ = forward

    # -- fused fast paths -----------------------------------------------------------------------
    def greedy(self, input, widths=None):
        """Forward + greedy CTC collapse on the device (test.py:191-194 + utils/ctc_codec.py:70-99).
        Returns a list of int32 label arrays, one per line."""
        ctx = self._require_ctx()
        x, dt, on_dev, B, W = self._img_args(input)
        wd = self._widths(widths, B)
        labels = np.empty((B, W), dtype=np.int32)
        lengths = np.empty((B,), dtype=np.int32)
        _lib.check(_lib.load().hctr_greedy(ctx, _lib.ptr(x), dt, on_dev, _lib.ptr(wd), B, W, _lib.ptr(labels),
                                           _lib.ptr(lengths)), ctx)
        return [labels[b, :lengths[b]].copy() for b in range(B)]

    def beam_frontend(self, input, k, widths=None, want_candidates=False):
        """Forward + log-softmax + top-k (+ thresholded candidate lists) on the device.
        Returns a dict consumed by ``ctc_codec.decode_frontend``."""
        ctx = self._require_ctx()
        x, dt, on_dev, B, W = self._img_args(input)
        wd = self._widths(widths, B)
        return beam_frontend_call(ctx, x, dt, on_dev, wd, None, 0, B, W, int(self.noutput), k, want_candidates)

    # -- precision mode ---------------------------------------------------------------------------
    def set_precision(self, precision):
        """Switch the mode of a loaded model among those whose weight set is resident (all three for a model built
        with precision="auto"; raises RuntimeError otherwise)."""
        if precision not in _PRECISIONS:
            raise ValueError("precision must be 'f16', 'f16x3' or 'auto'")
        if self._ctx is not None:
            _lib.check(_lib.load().hctr_set_precision(self._ctx, _PRECISIONS[precision]), self._ctx)
            self._active_precision = precision
        else:
            self.precision = precision
        return self

    def set_guard(self, rel=0.01, abs=0.05):
        """Criterion of the "auto" mode: a line is run again in f16x3 unless every column's top-1/top-2 logit margin
        exceeds 2 * (rel * max|logit of the line| + abs)."""
        _lib.check(_lib.load().hctr_set_guard(self._require_ctx(), float(rel), float(abs)), self._ctx)
        return self

    def last_guard(self):
        """Figures of the last call in "auto" mode: dict(lines, flagged, flags[uint8], min_margin[float32],
        scale[float32]) per line of that call's batch (empty arrays after a call in another mode)."""
        ctx = self._require_ctx()
        lib = _lib.load()
        n, nf = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(lib.hctr_last_guard(ctx, ctypes.byref(n), ctypes.byref(nf), None, None, None, 0), ctx)
        flags = np.zeros((n.value,), np.uint8)
        mg = np.zeros((n.value,), np.float32)
        sc = np.zeros((n.value,), np.float32)
        _lib.check(lib.hctr_last_guard(ctx, None, None, _lib.ptr(flags), _lib.ptr(mg), _lib.ptr(sc), n.value), ctx)
        return {"lines": int(n.value), "flagged": int(nf.value), "flags": flags, "min_margin": mg, "scale": sc}

    def lines_per_pass(self, B, W, f16x3=None):
        """Lines per internal pass of a batch of B lines of width W (HCTR_MAX_COLS pixel columns, a third in f16x3)."""
        if f16x3 is None:
            f16x3 = getattr(self, "_active_precision", self.precision) == "f16x3"
        n = _lib.load().hctr_lines_per_pass(self._require_ctx(), int(B), int(W), int(bool(f16x3)))
        if n < 0:
            _lib.check(n, self._ctx)
        return n

    # -- introspection --------------------------------------------------------------------------
    def workspace_stats(self):
        """dict(arena_bytes, arena_allocations, recarves) of the engine's workspace arena."""
        a, n, r = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(_lib.load().hctr_workspace_stats(self._require_ctx(), ctypes.byref(a), ctypes.byref(n), ctypes.byref(r)),
                   self._ctx)
        return {"arena_bytes": int(a.value), "arena_allocations": int(n.value), "recarves": int(r.value)}

    def set_profiling(self, enabled=True):
        _lib.check(_lib.load().hctr_set_profiling(self._require_ctx(), int(enabled)), self._ctx)

    def last_profile(self):
        """[(layer name, milliseconds)] of the last forward (HIP events on the engine's stream)."""
        ctx = self._require_ctx()
        names = ctypes.create_string_buffer(1 << 14)
        ms = (ctypes.c_float * 256)()
        n = _lib.load().hctr_last_profile(ctx, names, len(names), ms, 256)
        if n < 0:
            _lib.check(n, ctx)
        nm = names.value.decode().split("\n")
        return [(nm[i], float(ms[i])) for i in range(n)]

    def debug_activation(self, name, batch):
        """float32 NCHW copy of an intermediate activation of the last forward (parity bisecting).
        ``batch`` is the batch size of that forward."""
        ctx = self._require_ctx()
        lib = _lib.load()
        C, H = ctypes.c_int(), ctypes.c_int()
        n = lib.hctr_debug_activation(ctx, name.encode(), None, 0, ctypes.byref(C), ctypes.byref(H))
        if n < 0:
            _lib.check(int(n), ctx)
        out = np.empty((n,), dtype=np.float32)
        n2 = lib.hctr_debug_activation(ctx, name.encode(), _lib.ptr(out), n, ctypes.byref(C), ctypes.byref(H))
        if n2 < 0:
            _lib.check(int(n2), ctx)
        return out.reshape(batch, C.value, H.value, -1)


def beam_frontend_call(ctx, x, dt, on_dev, widths, logits, logits_on_dev, B, W, C, k, want_candidates):
    lib = _lib.load()
    topk_idx = np.empty((W, B, k), dtype=np.int32)
    topk_logp = np.empty((W, B, k), dtype=np.float32)
    blank = np.empty((W, B), dtype=np.float32)
    ncand = ctypes.c_int64(0)
    _lib.check(lib.hctr_beam_frontend(ctx, _lib.ptr(x), dt, on_dev, _lib.ptr(widths), _lib.ptr(logits), logits_on_dev,
                                      B, W, C, k, int(bool(want_candidates)), _lib.ptr(topk_idx), _lib.ptr(topk_logp),
                                      _lib.ptr(blank), ctypes.byref(ncand)), ctx)
    fe = {"W": W, "B": B, "C": C, "k": k, "topk_idx": topk_idx, "topk_logp": topk_logp, "blank_logp": blank,
          "cand_off": None, "cand_idx": None, "cand_logp": None}
    if want_candidates:
        fe["cand_off"] = np.zeros((W * B + 1,), dtype=np.int64)
        fe["cand_idx"] = np.empty((max(1, ncand.value),), dtype=np.int32)
        fe["cand_logp"] = np.empty((max(1, ncand.value),), dtype=np.float32)
        if W * B > 0:
            _lib.check(lib.hctr_beam_fetch_candidates(ctx, _lib.ptr(fe["cand_off"]), _lib.ptr(fe["cand_idx"]),
                                                      _lib.ptr(fe["cand_logp"])), ctx)
    return fe
