"""The product's hit-list sorts (csrc/rtow_sample_kernel.hip.h, between the "[hit-list sorts: ...]" markers) compiled for the host from
the very same text, under UBSan, against the oracle's NativeSortExtension restatement: every list length around the 24-entry boundary of
the spilled layout, tie-heavy keys, and the adversarial inputs that reach the introsort's heap-sort fallback.  The device build of the
same text is checked by tests/test_gpu_hitsort.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP = C.POINTER(C.c_float)
IP = C.POINTER(C.c_int)


@pytest.fixture(scope="module")
def hostsort():
    hdr = os.path.join(ROOT, "raytracing-in-one-weekend_amd", "csrc", "rtow_sample_kernel.hip.h")
    text = open(hdr).read()
    block = text[text.index("// [hit-list sorts: begin]"):text.index("// [hit-list sorts: end]")]
    out_dir = os.path.join(ROOT, "tests", "build")
    os.makedirs(out_dir, exist_ok=True)
    inc = os.path.join(out_dir, "hitsort_extracted.inc")
    if not os.path.exists(inc) or open(inc).read() != block:
        open(inc, "w").write(block)
    so = os.path.join(out_dir, "libhitsort_host.so")
    src = os.path.join(ROOT, "tests", "native", "hitsort_host.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(inc)):
        subprocess.run(["g++", "-std=c++17", "-O2", "-g", "-fsanitize=undefined", "-fno-sanitize-recover=all", "-fPIC", "-shared", src, "-o", so],
                       check=True, capture_output=True)
    lib = C.CDLL(so)
    lib.hitsort_host_run.argtypes = [FP, C.c_int, IP]
    lib.hitsort_host_run.restype = C.c_int
    return lib


def _both(hostsort, keys):
    keys = np.ascontiguousarray(keys, np.float32)
    n = len(keys)
    got = np.empty(n, np.int32)
    assert hostsort.hitsort_host_run(keys.ctypes.data_as(FP), n, got.ctypes.data_as(IP)) == 0
    k, want = keys.copy(), np.arange(n, dtype=np.int32)
    ob.load().oracle_kat_unity_sort(k.ctypes.data_as(FP), want.ctypes.data_as(IP), n)
    return got, want


def test_host_build_of_the_device_sorts_equals_the_oracle(hostsort):
    rng = np.random.default_rng(7)
    for n in list(range(1, 70)) + [97, 128, 200, 513, 1000]:
        cases = [np.arange(n), np.arange(n)[::-1], np.zeros(n), np.arange(n) // 2, (np.arange(n) * 7) % 5]
        for distinct in (2, 3, max(2, n // 4), 4 * n):
            cases += [rng.integers(0, distinct, n) for _ in range(12)]
        for keys in cases:
            got, want = _both(hostsort, keys)
            assert got.tolist() == want.tolist(), (n, np.asarray(keys).tolist())


def test_host_build_reaches_the_heap_sort_fallback(hostsort):
    lib = ob.load()
    for n in (40, 64, 100, 333, 1000):
        killer = np.zeros(n, np.float32)
        lib.oracle_kat_unity_sort_killer(n, killer.ctypes.data_as(FP))
        lib.oracle_kat_unity_sort_heapsorts()
        got, want = _both(hostsort, killer)
        assert lib.oracle_kat_unity_sort_heapsorts() >= 1
        assert got.tolist() == want.tolist(), n
        for fold in (2, 3, 5):
            got, want = _both(hostsort, np.floor(killer / fold))
            assert got.tolist() == want.tolist(), (n, fold)
