"""Build and bind libhctr_hip.so (the C ABI in include/hctr_hip.h) with ctypes.

The shared library is built IN-TREE with hipcc for gfx950 (it travels to the GPU box with the
snapshot). There is no fallback: if the library is missing or fails to load, importing callers get a
loud error - the engine has no CPU path.
"""
import ctypes
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libhctr_hip.so")
HEADER = os.path.join(ROOT, "include", "hctr_hip.h")

# (source, extra flags). beam_search.cpp must not contract a*b+c (bit-parity with Python floats).
SOURCES = [("gather.cpp", ["-x", "hip"]), ("kernels.hip", []), ("preprocess.hip", ["-ffp-contract=off"]), ("engine.cpp", ["-x", "hip"]), ("beam_search.cpp", ["-ffp-contract=off"]),
           ("ngram_lm.cpp", ["-ffp-contract=off"])]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]

HCTR_OK = 0
ERR_ARG, ERR_HIP, ERR_STATE, ERR_KEY, ERR_SHAPE, ERR_EMPTY_LINE, ERR_NOMEM = -1, -2, -3, -4, -5, -6, -7
U8, F32, I64 = 0, 1, 2


def _deps():
    return [os.path.join(CSRC, s) for s, _ in SOURCES] + [os.path.join(CSRC, "kernels.h"),
                                                          os.path.join(CSRC, "ngram_lm.h"), HEADER]


def source_hash():
    """16 hex digits over the contents of every source the library is built from and the compile flags.
    Compiled into the library (``hctr_version()`` ends in ``src=<hash>``), so "is this binary built from these
    sources" does not depend on file times (a snapshot copied to the GPU box keeps no useful mtimes)."""
    import hashlib
    h = hashlib.sha256()
    h.update(repr((COMMON, SOURCES)).encode())
    for d in sorted(_deps()):
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


_MARK = b"hctr-src="


def embedded_hash(path=None):
    """The source hash compiled into a built library, read from the file (None if absent)."""
    path = path or LIB_PATH
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    i = blob.find(_MARK)
    if i < 0:
        return None
    return blob[i + len(_MARK):i + len(_MARK) + 16].decode("ascii", "replace")


def _stale():
    return embedded_hash() != source_hash()


def build(force=False, verbose=False):
    """Compile csrc/ into libhctr_hip.so (cross-compiles without a GPU). Safe to call from several
    processes at once (one rank per GPU): an exclusive file lock serialises the build, later callers find
    the library fresh, and the link goes to a temporary name that is renamed into place."""
    if not force and not _stale():
        return LIB_PATH
    import fcntl
    os.makedirs(os.path.join(PKG_DIR, "build"), exist_ok=True)
    with open(os.path.join(PKG_DIR, "build", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return LIB_PATH
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.isfile(hipcc):
        hipcc = "hipcc"
    objdir = os.path.join(PKG_DIR, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    srch = source_hash()
    for src, extra in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        if src == "engine.cpp":
            extra = extra + ['-DHCTR_SRC_HASH="%s"' % srch]
        cmd = [hipcc] + COMMON + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        objs.append(obj)
        # an object is reused when its source, the headers and its command line are unchanged
        import hashlib
        h = hashlib.sha256(repr(cmd).encode())
        for d in [os.path.join(CSRC, src), os.path.join(CSRC, "kernels.h"), os.path.join(CSRC, "ngram_lm.h"), HEADER]:
            with open(d, "rb") as f:
                h.update(f.read())
        tag, tagfile = h.hexdigest(), obj + ".tag"
        if os.path.isfile(obj) and os.path.isfile(tagfile) and open(tagfile).read() == tag:
            continue
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        with open(tagfile, "w") as f:
            f.write(tag)
    tmp = LIB_PATH + ".tmp.%d" % os.getpid()
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + ["-lpthread", "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB_PATH)
    if embedded_hash() != srch:
        raise OSError("built library does not carry the source hash %s" % srch)
    return LIB_PATH


_LIB = None

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)

LM_SCORE_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, c_i32p, c_i32p, c_f64p)
LM_NEXT_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, c_i32p, c_i32p, ctypes.c_int, c_i32p)


class BeamParams(ctypes.Structure):
    _fields_ = [("skip_search", ctypes.c_int), ("beam_size", ctypes.c_int), ("search_depth", ctypes.c_int),
                ("lm_panelty", ctypes.c_double), ("len_bonus", ctypes.c_double),
                ("builtin_lm", ctypes.c_int), ("label_codepoints", ctypes.c_void_p),
                ("score_cb", LM_SCORE_CB), ("next_cb", LM_NEXT_CB), ("user", ctypes.c_void_p),
                ("num_threads", ctypes.c_int), ("ngram", ctypes.c_void_p), ("label_words", ctypes.c_void_p)]


# every symbol include/hctr_hip.h declares: (name, restype, argtypes)
_VP, _I, _I64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
SIGNATURES = [
    ("hctr_create", _I, [ctypes.POINTER(_VP), _I, _I]),
    ("hctr_destroy", None, [_VP]),
    ("hctr_last_error", ctypes.c_char_p, [_VP]),
    ("hctr_version", ctypes.c_char_p, []),
    ("hctr_load_tensor", _I, [_VP, ctypes.c_char_p, _VP, c_i64p, _I, _I]),
    ("hctr_finalize_weights", _I, [_VP]),
    ("hctr_set_precision", _I, [_VP, _I]),
    ("hctr_set_guard", _I, [_VP, ctypes.c_double, ctypes.c_double]),
    ("hctr_last_guard", _I, [_VP, c_i64p, c_i64p, _VP, _VP, _VP, _I64]),
    ("hctr_lines_per_pass", _I, [_VP, _I, _I, _I]),
    ("hctr_workspace_stats", _I, [_VP, c_i64p, c_i64p, c_i64p]),
    ("hctr_forward_logits", _I, [_VP, _VP, _I, _I, _VP, _I, _I, _VP, _I]),
    ("hctr_greedy", _I, [_VP, _VP, _I, _I, _VP, _I, _I, _VP, _VP]),
    ("hctr_decode_greedy_logits", _I, [_VP, _VP, _I, _I, _I, _I, _VP, _VP]),
    ("hctr_beam_frontend", _I, [_VP, _VP, _I, _I, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP, c_i64p]),
    ("hctr_beam_fetch_candidates", _I, [_VP, _VP, _VP, _VP]),
    ("hctr_log_softmax", _I, [_VP, _VP, _I, _I, _I, _I, _VP]),
    ("hctr_beam_search", _I, [ctypes.POINTER(BeamParams), _I, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                              _VP, _VP, _VP]),
    ("hctr_ngram_load", _I, [ctypes.c_char_p, ctypes.POINTER(_VP)]),
    ("hctr_ngram_free", None, [_VP]),
    ("hctr_ngram_order", _I, [_VP]),
    ("hctr_ngram_word_id", ctypes.c_int32, [_VP, ctypes.c_char_p]),
    ("hctr_ngram_score", ctypes.c_double, [_VP, ctypes.c_char_p, _I, _I]),
    ("hctr_ngram_last_error", ctypes.c_char_p, []),
    ("hctr_comm_unique_id", _I, [_VP]),
    ("hctr_comm_create", _I, [ctypes.POINTER(_VP), _VP, _I, _I, _I]),
    ("hctr_comm_destroy", None, [_VP]),
    ("hctr_gather_labels", _I, [_VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP]),
    ("hctr_comm_last_error", ctypes.c_char_p, []),
    ("hctr_resize_lines", _I, [_VP, _VP, _I64, c_i64p, c_i32p, c_i32p, c_i32p, _I, _I, c_i32p, _I, _VP, _I]),
    ("hctr_set_profiling", _I, [_VP, _I]),
    ("hctr_last_profile", _I, [_VP, ctypes.c_char_p, _I, c_f32p, _I]),
    ("hctr_debug_stamps", _I64, [_VP, ctypes.c_char_p, _VP, _I64]),
    ("hctr_debug_activation", _I64, [_VP, ctypes.c_char_p, _VP, _I64, ctypes.POINTER(_I), ctypes.POINTER(_I)]),
]


def load():
    """Load the library (building it first when the sources are newer). Raises on failure."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # HCTR_LIB_PATH: load an alternative build of the library as it is (same-box A/B of compile-time switches)
    alt = os.environ.get("HCTR_LIB_PATH", "")
    if alt:
        if not os.path.isfile(alt):
            raise RuntimeError("HCTR_LIB_PATH=%s does not exist" % alt)
    elif _stale():
        # A failed rebuild is an error: silently loading the previous binary would run tests and benches against
        # code that no longer matches the sources. The one exception is a machine without hipcc (FileNotFoundError)
        # that was shipped a prebuilt library - then that library is used, with a warning.
        try:
            build()
        except FileNotFoundError as exc:
            if not os.path.isfile(LIB_PATH):
                raise RuntimeError("libhctr_hip.so is missing and hipcc is not available (%s); "
                                   "the hctr engine has no CPU fallback" % exc)
            print("hctr: WARNING: sources are newer than libhctr_hip.so but hipcc is not available; "
                  "loading the prebuilt library", file=sys.stderr)
        except (OSError, subprocess.CalledProcessError) as exc:
            raise RuntimeError("libhctr_hip.so is out of date and rebuilding it failed (%s); fix the build "
                               "(python -c 'import hctr_amd; hctr_amd.build(force=True, verbose=True)')" % exc)
    # One HIP runtime per process: PyTorch ships its own libamdhip64.so.7 (same SONAME as /opt/rocm's).
    # Loaded first, it is the copy this library binds to as well, so device pointers of torch tensors
    # (on_device arguments, DLPack-free interop) and the engine's own allocations share a runtime; loaded
    # second, torch would bring up a second runtime that finds no GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(alt or LIB_PATH)
    if not alt and embedded_hash() != source_hash():
        print("hctr: WARNING: libhctr_hip.so was built from other sources than csrc/ holds now (%s vs %s)"
              % (embedded_hash(), source_hash()), file=sys.stderr)
    # HCTR_HOST_ONLY=1 (with HCTR_LIB_PATH): a sanitizer build of the pure-host sources only (beam search, n-gram
    # scorer; tools/build_host_sanitized.sh) - the device entry points are then absent and stay unbound
    host_only = bool(alt) and os.environ.get("HCTR_HOST_ONLY", "") == "1"
    for name, res, args in SIGNATURES:
        if host_only and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)          # AttributeError if the ABI drifted from the header
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


_EXC = {ERR_ARG: ValueError, ERR_HIP: RuntimeError, ERR_STATE: RuntimeError, ERR_KEY: KeyError,
        ERR_SHAPE: RuntimeError, ERR_EMPTY_LINE: IndexError, ERR_NOMEM: MemoryError}


def check(rc, ctx=None):
    """Turn a negative hctr_status into the exception type the reference would raise."""
    if rc == HCTR_OK:
        return
    msg = load().hctr_last_error(ctx)
    msg = msg.decode("utf-8", "replace") if msg else ""
    raise _EXC.get(rc, RuntimeError)("hctr engine error %d: %s" % (rc, msg))


def usable_cpus():
    """CPUs this process can really keep busy: its affinity mask, capped by the cgroup CPU quota when there is one
    (a container with cpu.max = 16 CPUs on a 256-thread host is throttled by the scheduler if 64 threads spin)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            with open(path) as f:
                q, per = f.read().split()
            if q != "max":
                n = min(n, max(1, int(float(q) / float(per) + 0.5)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def ptr(a):
    """ctypes void* of a numpy array / torch tensor / None."""
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return ctypes.c_void_p(a.data_ptr())
    return a.ctypes.data_as(ctypes.c_void_p)
