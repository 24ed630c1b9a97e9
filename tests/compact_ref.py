"""numpy restatement of the accept-compacted wire format (include/mhx.h: mhx_compact_hdr) -- TEST INFRASTRUCTURE.

`encode_block` does on the host what the three device kernels of advancedmh.jl_amd/csrc/mhx_api_host.inc do to one slab: a bit per
(sample, chain) "the column differs from the sample before" (compared as BITS, so NaN payloads and signed zeros count), the
exclusive ranks of the mask words, and the changed columns parameter-major.  `decode_block` is the straight-line reading of the
format that mhx_compact_expand (host threads, AVX-512 expand-loads) must agree with."""
import ctypes as C

import numpy as np


def _bits(a):
    return a.view({4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


def pad8(x):
    return (x + 7) & ~7


def encode_block(tensor, accepted, first, count, changed=None):
    """tensor [N][d1][n] (float32 / float64), accepted [N][n] uint8 -> bytes of the block for samples [first, first + count).
    `changed` [count][n] bool overrides the comparison (a superset of the truly changed chains is a legal block)."""
    from mhx import _lib as L
    N, d1, n = tensor.shape
    eb = tensor.dtype.itemsize
    words = (n + 63) // 64
    blk = tensor[first:first + count]
    if changed is None:
        if first == 0:
            above = np.concatenate([blk[:1], blk[:-1]])           # row 0 compares with itself, then forced below
        else:
            above = tensor[first - 1:first + count - 1]
        changed = (_bits(blk) != _bits(above)).any(axis=1)
        if first == 0:
            changed[0] = True
    changed = np.asarray(changed, dtype=bool)
    padded = np.zeros((count, words * 64), dtype=bool)
    padded[:, :n] = changed
    # bit c % 64 of word c // 64
    mask = (padded.reshape(count, words, 64).astype(np.uint64) << np.arange(64, dtype=np.uint64)).sum(axis=2, dtype=np.uint64)
    per_word = padded.reshape(count, words, 64).sum(axis=2).astype(np.uint64).ravel()
    rank = np.concatenate([[0], np.cumsum(per_word)[:-1]]).astype(np.uint32)
    total = int(per_word.sum())
    payload = np.concatenate([blk[i][:, changed[i]].ravel() for i in range(count)]) if total else np.empty(0, tensor.dtype)
    nw = count * words
    po = 64 + 8 * nw + pad8(4 * nw) + pad8(count * n)
    h = L.CompactHdr(L.COMPACT_MAGIC, eb, d1, n, first, count, words, total, po, po + total * d1 * eb)
    buf = bytearray(po + total * d1 * eb)
    buf[0:64] = bytes(h)
    buf[64:64 + 8 * nw] = mask.tobytes()
    buf[64 + 8 * nw:64 + 12 * nw] = rank.tobytes()
    a0 = 64 + 8 * nw + pad8(4 * nw)
    buf[a0:a0 + count * n] = np.ascontiguousarray(accepted[first:first + count]).tobytes()
    buf[po:] = np.ascontiguousarray(payload).tobytes()
    return bytes(buf)


def decode_block(block, tensor, accepted):
    """the format read literally, one chain at a time (slow; small cases)"""
    from mhx import _lib as L
    h = L.CompactHdr.from_buffer_copy(block[:64])
    dt = {4: np.float32, 8: np.float64}[h.elem_bytes]
    nw = h.count * h.words
    mask = np.frombuffer(block, np.uint64, nw, 64).reshape(h.count, h.words)
    rank = np.frombuffer(block, np.uint32, nw, 64 + 8 * nw).reshape(h.count, h.words)
    a0 = 64 + 8 * nw + pad8(4 * nw)
    acc = np.frombuffer(block, np.uint8, h.count * h.nchains, a0).reshape(h.count, h.nchains)
    pay = np.frombuffer(block, dt, h.total_changed * h.dim1, h.payload_offset)
    for i in range(h.count):
        row = h.first_sample + i
        if row:
            tensor[row] = tensor[row - 1]
        first = int(rank[i, 0])
        m = (int(rank[i + 1, 0]) if i + 1 < h.count else h.total_changed) - first
        for c in range(h.nchains):
            w, b = divmod(c, 64)
            bits = int(mask[i, w])
            if (bits >> b) & 1:
                r = int(rank[i, w]) - first + bin(bits & ((1 << b) - 1)).count("1")
                tensor[row, :, c] = pay[first * h.dim1 + np.arange(h.dim1) * m + r]
        if accepted is not None:
            accepted[row] = acc[i]


def expand(block, tensor, accepted, threads=0):
    from mhx import _lib as L
    L.check(L.lib().mhx_compact_expand(block, len(block), L.rptr(tensor), L.u8ptr(accepted), tensor.shape[0], threads))


def synthetic_chain(rng, N, d1, n, dtype, p_accept):
    """a tensor with the repeat structure of a Metropolis chain: a chain's column changes with probability p_accept"""
    t = np.empty((N, d1, n), dtype=dtype)
    acc = (rng.random((N, n)) < p_accept).astype(np.uint8)
    t[0] = rng.standard_normal((d1, n))
    for i in range(1, N):
        t[i] = t[i - 1]
        m = acc[i].astype(bool)
        t[i][:, m] = rng.standard_normal((d1, int(m.sum())))
    return t, acc
