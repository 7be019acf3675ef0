"""-m gpu: the hand-written stable LSD radix sort vs numpy's stable sort."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import pyivx  # noqa: E402

pytestmark = pytest.mark.gpu


def _sort(ctx, words, fields):
    n = len(words[0])
    arrs = [np.ascontiguousarray(w, np.uint64).copy() for w in words] + [None] * (3 - len(words))
    fw = (C.c_int * len(fields))(*[f[0] for f in fields])
    fl = (C.c_int * len(fields))(*[f[1] for f in fields])
    fh = (C.c_int * len(fields))(*[f[2] for f in fields])
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    st = pyivx.lib().ivx_debug_sort(ctx.h, C.c_int(len(words)), p(arrs[0]), p(arrs[1]), p(arrs[2]), C.c_uint64(n),
                                    fw, fl, fh, C.c_int(len(fields)))
    assert st == 0, pyivx.lib().ivx_last_error(ctx.h)
    return arrs[:len(words)]


def _ref(words, fields):
    order = np.arange(len(words[0]))
    for w, lo, hi in fields:                      # LSD: least significant criterion first, stable
        mask = (np.uint64(1) << np.uint64(hi - lo)) - np.uint64(1) if hi - lo < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
        k = (words[w][order] >> np.uint64(lo)) & mask
        order = order[np.argsort(k, kind="stable")]
    return [w[order] for w in words]


@pytest.mark.parametrize("n", [0, 1, 63, 2049, 16384, 16385, 300_000, 2_000_003])
def test_sort_two_words(n):
    ctx = pyivx.Ctx(0)
    rng = np.random.default_rng(n)
    w0 = rng.integers(0, 1 << 63, n, dtype=np.uint64) & np.uint64(0x0000FFFF0000FFFF)   # many ties, constant bytes
    w1 = (rng.integers(0, 24, n, dtype=np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    fields = [(0, 0, 32), (0, 32, 64), (1, 32, 64)]
    got = _sort(ctx, [w0, w1], fields)
    want = _ref([w0, w1], fields)
    for g, w in zip(got, want):
        assert (g == w).all()
    ctx.close()


def test_sort_three_words_full_range():
    ctx = pyivx.Ctx(0)
    rng = np.random.default_rng(7)
    n = 777_777
    w0 = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    w1 = rng.integers(0, 1 << 20, n, dtype=np.uint64)                      # heavy duplicates -> stability matters
    w2 = (rng.integers(0, 300, n, dtype=np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    fields = [(0, 0, 64), (1, 0, 64), (2, 32, 64)]
    got = _sort(ctx, [w0, w1, w2], fields)
    want = _ref([w0, w1, w2], fields)
    for g, w in zip(got, want):
        assert (g == w).all()
    ctx.close()


def test_sort_skewed_single_digit():
    ctx = pyivx.Ctx(0)
    n = 500_000
    w0 = np.full(n, 5, np.uint64); w0[::1000] = 3; w0[123] = 255
    w1 = np.arange(n, dtype=np.uint64)
    got = _sort(ctx, [w0, w1], [(0, 0, 8)])
    want = _ref([w0, w1], [(0, 0, 8)])
    assert (got[0] == want[0]).all() and (got[1] == want[1]).all()
    ctx.close()


@pytest.mark.parametrize("n,lo,hi,pay,tight", [(1, 0, 8, True, 0), (8191, 26, 58, True, 1), (8193, 26, 58, False, 1), (70_001, 3, 36, True, 0),
                                              (1_000_003, 26, 58, True, 1), (1_000_003, 30, 63, False, 0), (3_000_000, 20, 53, True, 1)])
def test_sort_one_word_with_payload(n, lo, hi, pay, tight):
    # 8-byte records (+ a 32-bit payload riding along): the form the nearest index build and the sweeps sort
    ctx = pyivx.Ctx(0)
    rng = np.random.default_rng(n + lo)
    w = rng.integers(0, 1 << 63, n, dtype=np.uint64)
    if hi < 64:
        w &= (np.uint64(1) << np.uint64(hi)) - np.uint64(1)
    w = (w >> np.uint64(lo) << np.uint64(lo)) | (np.arange(n, dtype=np.uint64) & ((np.uint64(1) << np.uint64(lo)) - np.uint64(1)))
    if n > 8:
        w[: n // 7 * 7 : 7] = w[1 : n // 7 * 7 : 7]                   # ties
    p = rng.integers(0, 1 << 32, n, dtype=np.uint32)
    gw, gp = w.copy(), p.copy()
    st = pyivx.lib().ivx_debug_sort_pay(ctx.h, gw.ctypes.data_as(C.c_void_p), gp.ctypes.data_as(C.c_void_p) if pay else None,
                                        C.c_uint64(n), C.c_int(lo), C.c_int(hi), C.c_int(tight))
    assert st == 0, pyivx.lib().ivx_last_error(ctx.h)
    mask = (np.uint64(1) << np.uint64(hi - lo)) - np.uint64(1)
    order = np.argsort((w >> np.uint64(lo)) & mask, kind="stable")
    assert (gw == w[order]).all()
    if pay:
        assert (gp == p[order]).all()
    ctx.close()
