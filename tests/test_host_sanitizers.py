"""Host-code sanitizers (the reference has none; GPU sanitizers are not available on this
pool, so this covers the pure-host C++ only): the prefix beam search and the ARPA n-gram scorer are rebuilt with
g++ -fsanitize=address,undefined and -fsanitize=thread (tools/build_host_sanitized.sh) and the existing CPU tests
of that code - reference goldens, callback LM, hooks, ARPA back-off cases, threaded built-in LMs, hypothesis fuzz -
run against the instrumented library in a child interpreter with the sanitizer runtime preloaded."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT


def _runtime(name):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    path = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(path) or not os.path.isfile(path):
        pytest.skip(name + " not available")
    return path


def _run(kind, runtime, select, extra_env):
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "build_host_sanitized.sh"), kind], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = r.stdout.strip().splitlines()[-1]
    env = dict(os.environ)
    env.update({"LD_PRELOAD": runtime, "HCTR_LIB_PATH": lib, "HCTR_HOST_ONLY": "1"})
    env.update(extra_env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_host_cpu.py"), "-x", "-q",
                        "-p", "no:cacheprovider", "-k", select], env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "passed" in r.stdout and "Sanitizer" not in out and "runtime error:" not in out, out[-3000:]


def test_beam_search_and_ngram_under_asan_ubsan():
    _run("asan", _runtime("libasan.so"), "beam or arpa",
         {"ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"})


def test_threaded_beam_search_under_tsan():
    _run("tsan", _runtime("libtsan.so"), "beam_search_matches_reference or native_arpa or beam_search_fuzz",
         {"TSAN_OPTIONS": "halt_on_error=1:report_signal_unsafe=0:exitcode=66"})
