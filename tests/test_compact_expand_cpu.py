"""CPU: the host half of the accept-compacted return path -- mhx_compact_expand (advancedmh.jl_amd/csrc/mhx_host_expand.cpp) rebuilds
the caller's tensor from blocks in the wire format of include/mhx.h.  Blocks are made by the numpy restatement of the device's
three kernels (tests/compact_ref.py); the result must equal the original tensor bit for bit, in both widths, at chain counts that
are not multiples of 64 / of the chunk, with one thread and many, with the AVX-512 rows and the scalar rows.  What it replaces:
the reference's `sample` returns one row per iteration, repeats included (src/mh-core.jl:109-114, ext/AdvancedMHMCMCChainsExt.jl:12-39).
No GPU is touched: the library loads and the function runs on the host."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import compact_ref as R


def _same(a, b):
    v = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
    return np.array_equal(a.view(v), b.view(v))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,d1,N,slab", [(1, 3, 9, 4), (63, 2, 7, 7), (64, 5, 12, 5), (130, 4, 10, 3), (1000, 7, 20, 6), (4097, 3, 11, 4),
                                         (70000, 3, 6, 2)])
@pytest.mark.parametrize("threads", [1, 5])
def test_blocks_expand_to_the_tensor(mhx, dtype, n, d1, N, slab, threads):
    rng = np.random.default_rng(n * 31 + d1)
    t, acc = R.synthetic_chain(rng, N, d1, n, dtype, 0.24)
    # payloads the comparison must not normalise: NaNs with different payload bits, signed zeros
    t[N // 2, 0, 0] = np.nan
    t[N // 2 + 1:, 0, 0] = -0.0
    out = np.full_like(t, 7.0)
    oacc = np.full_like(acc, 9)
    wire = 0
    for first in range(0, N, slab):
        blk = R.encode_block(t, acc, first, min(slab, N - first))
        wire += len(blk)
        R.expand(blk, out, oacc, threads)
    assert _same(out, t) and np.array_equal(oacc, acc)
    if n >= 1000:
        assert wire < 0.6 * t.nbytes                  # ~ acceptance + sample 0 whole + masks, ranks, accept flags


def test_the_literal_reading_of_the_format_agrees(mhx):
    rng = np.random.default_rng(5)
    t, acc = R.synthetic_chain(rng, 8, 4, 150, np.float64, 0.3)
    a, b = np.zeros_like(t), np.zeros_like(t)
    aa, ba = np.zeros_like(acc), np.zeros_like(acc)
    for first in (0, 3, 6):
        blk = R.encode_block(t, acc, first, min(3, 8 - first))
        R.decode_block(blk, a, aa)
        R.expand(blk, b, ba, 2)
    assert _same(a, t) and _same(b, t) and np.array_equal(aa, acc) and np.array_equal(ba, acc)


def test_a_superset_of_the_changed_chains_is_a_legal_block(mhx):
    """thinning > 1 or a sampler that re-emits an equal state: sending an unchanged column is harmless"""
    rng = np.random.default_rng(8)
    t, acc = R.synthetic_chain(rng, 6, 3, 200, np.float32, 0.2)
    out = np.zeros_like(t)
    for first in (0, 2, 4):
        ch = rng.random((2, 200)) < 0.6
        true = (t[first:first + 2].view(np.uint32) != t[np.maximum(np.arange(first, first + 2) - 1, 0)].view(np.uint32)).any(axis=1)
        ch |= true
        if first == 0:
            ch[0] = True
        R.expand(R.encode_block(t, acc, first, 2, changed=ch), out, None, 3)
    assert _same(out, t)


def test_all_changed_and_none_changed(mhx):
    rng = np.random.default_rng(1)
    n, d1, N = 256, 3, 5
    t = rng.standard_normal((N, d1, n))                          # every column new in every row
    acc = np.ones((N, n), np.uint8)
    out = np.zeros_like(t)
    R.expand(R.encode_block(t, acc, 0, N), out, None, 2)
    assert _same(out, t)
    t2 = np.repeat(t[:1], N, axis=0)                             # nothing ever moves
    blk = R.encode_block(t2, acc, 0, N)
    assert len(blk) < 1.3 * t2[0].nbytes + 64 + N * (n + 12 * 4) + 64
    out2 = np.zeros_like(t2)
    R.expand(blk, out2, None, 2)
    assert _same(out2, t2)


def test_malformed_blocks_are_refused_before_anything_is_written(mhx):
    rng = np.random.default_rng(2)
    t, acc = R.synthetic_chain(rng, 4, 3, 100, np.float64, 0.5)
    good = R.encode_block(t, acc, 0, 4)
    L = mhx._lib if hasattr(mhx, "_lib") else __import__("mhx._lib", fromlist=["x"])
    out = np.full_like(t, 3.0)

    def refuse(b, nbytes=None, rows=4):
        rc = L.lib().mhx_compact_expand(b, len(b) if nbytes is None else nbytes, L.rptr(out), None, rows, 1)
        assert rc == L.MHX_EINVAL, L.lib().mhx_last_error()
        assert (out == 3.0).all()

    refuse(b"\0" * 64)                                           # no magic
    refuse(good, nbytes=32)                                      # no header
    refuse(good, nbytes=len(good) - 8)                           # truncated payload
    refuse(good, rows=3)                                         # more samples than the tensor has rows
    bad = bytearray(good)
    h = L.CompactHdr.from_buffer(bad)
    nw = h.count * h.words
    r = np.frombuffer(bad, np.uint32, nw, 64 + 8 * nw)
    r[3] += 1                                                    # a rank that is not the running popcount
    refuse(bytes(bad))
    bad = bytearray(good)
    m = np.frombuffer(bad, np.uint64, nw, 64)
    m[1] |= np.uint64(1) << np.uint64(40)                        # chain 104 of 100
    refuse(bytes(bad))
    bad = bytearray(good)
    m = np.frombuffer(bad, np.uint64, nw, 64)
    m[0] &= ~np.uint64(1)                                        # sample 0 must carry every chain
    refuse(bytes(bad))
    bad = bytearray(good)
    L.CompactHdr.from_buffer(bad).elem_bytes = 2
    refuse(bytes(bad))
    R.expand(good, out, None, 1)                                 # and the good one still goes through
    assert _same(out, t)


def test_scalar_rows_equal_the_wide_rows():
    """MHX_EXPAND_NO_AVX512=1 takes the scalar merge + SSE2 streaming rows (hosts without AVX-512); same bytes.  The switch is read
    once per process, so the other form runs in a child."""
    code = r"""
import sys, numpy as np
sys.path[:0] = [%r, %r]
import compact_ref as R, mhx
rng = np.random.default_rng(11)
t, acc = R.synthetic_chain(rng, 9, 5, 3000, np.float64, 0.25)
t32, acc32 = R.synthetic_chain(rng, 9, 5, 3000, np.float32, 0.25)
for tt, aa in ((t, acc), (t32, acc32)):
    out = np.zeros_like(tt)
    for first in (0, 4, 8):
        R.expand(R.encode_block(tt, aa, first, min(4, 9 - first)), out, None, 3)
    v = {4: np.uint32, 8: np.uint64}[tt.dtype.itemsize]
    assert np.array_equal(out.view(v), tt.view(v))
print("ok")
""" % (os.path.dirname(os.path.abspath(__file__)), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "advancedmh.jl_amd"))
    for env in ({}, {"MHX_EXPAND_NO_AVX512": "1"}):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and "ok" in p.stdout, p.stderr[-2000:]


def test_usable_cpus_and_large_rows_stream(mhx):
    """a tensor past the streaming threshold (non-temporal rows), pageable destination at an odd byte offset"""
    rng = np.random.default_rng(3)
    n, d1, N = 8192, 20, 16                                      # 21 MB in fp64
    t, acc = R.synthetic_chain(rng, N, d1, n, np.float64, 0.234)
    raw = np.zeros(t.nbytes + 64, np.uint8)
    out = raw[8:8 + t.nbytes].view(np.float64).reshape(t.shape)  # 8-byte aligned only: the rows fall back to plain stores
    for first in range(0, N, 5):
        R.expand(R.encode_block(t, acc, first, min(5, N - first)), out, None, 0)
    assert _same(out, t)
    out2 = np.zeros_like(t)
    for first in range(0, N, 5):
        R.expand(R.encode_block(t, acc, first, min(5, N - first)), out2, None, 0)
    assert _same(out2, t)
