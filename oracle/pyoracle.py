"""ctypes binding for the CPU oracle (oracle/banzai_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (banzai_amd) never imports this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbanzai_oracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("banzai_oracle.c", "bz2_decode.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libbanzai_oracle.so"])
    return _SO


class BlockInfo(ctypes.Structure):
    _fields_ = [("in_off", ctypes.c_uint64), ("in_len", ctypes.c_uint64), ("rle_len", ctypes.c_uint64),
                ("m", ctypes.c_uint64), ("crc", ctypes.c_uint32), ("ptr", ctypes.c_uint32),
                ("num_syms", ctypes.c_uint32), ("pad", ctypes.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        u8p, u16p, u32p = (ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint16),
                           ctypes.POINTER(ctypes.c_uint32))
        szp = ctypes.POINTER(ctypes.c_size_t)
        L.orc_bitsink_run.restype = ctypes.c_size_t
        L.orc_bitsink_run.argtypes = [u32p, ctypes.c_size_t, u8p, u8p, ctypes.c_size_t]
        L.orc_crc32.restype = ctypes.c_uint32
        L.orc_crc32.argtypes = [u8p, ctypes.c_size_t]
        L.orc_rle_one.restype = ctypes.c_size_t
        L.orc_rle_one.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, u8p, szp, u32p]
        L.orc_bwt.restype = ctypes.c_size_t
        L.orc_bwt.argtypes = [u8p, ctypes.c_size_t, u8p, u8p]
        L.orc_bwt_naive.restype = ctypes.c_size_t
        L.orc_bwt_naive.argtypes = [u8p, ctypes.c_size_t, u8p]
        L.orc_mtf_and_rle.restype = ctypes.c_size_t
        L.orc_mtf_and_rle.argtypes = [u8p, ctypes.c_size_t, u8p, u16p, u32p, u32p]
        L.orc_build_table_from_freqs.restype = None
        L.orc_build_table_from_freqs.argtypes = [ctypes.c_uint32, u32p, u8p]
        L.orc_huffman_block.restype = ctypes.c_size_t
        L.orc_huffman_block.argtypes = [u16p, ctypes.c_size_t, ctypes.c_uint32, u32p, u8p, ctypes.c_size_t,
                                        u8p, u32p]
        L.orc_encode.restype = ctypes.c_size_t
        L.orc_encode.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, u8p, ctypes.c_size_t, szp,
                                 ctypes.POINTER(BlockInfo), ctypes.c_size_t, szp]
        L.orc_bz2_decode.restype = ctypes.c_int
        L.orc_bz2_decode.argtypes = [u8p, ctypes.c_size_t, u8p, ctypes.c_size_t, szp]
        _lib = L
    return _lib


def _u8(a):
    a = np.ascontiguousarray(np.frombuffer(bytes(a), dtype=np.uint8) if not isinstance(a, np.ndarray) else a,
                             dtype=np.uint8)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def bitsink_run(ops, blob=b""):
    """ops: list of (kind, value, nbits); see orc_bitsink_run."""
    arr = np.array(ops, dtype=np.uint32).reshape(-1)
    b, bp = _u8(blob if len(blob) else b"\0")
    out = np.zeros(4 * len(ops) + len(blob) + 8, dtype=np.uint8)
    n = lib().orc_bitsink_run(arr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), len(ops), bp,
                              out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), out.size)
    return out[:n].tobytes()


def crc32(data):
    a, p = _u8(data if len(data) else b"\0")
    return int(lib().orc_crc32(p, len(data)))


def rle_one(raw, level):
    """-> (rle_bytes, crc, consumed)"""
    a, p = _u8(raw if len(raw) else b"\0")
    out = np.zeros(100000 * level, dtype=np.uint8)
    olen = ctypes.c_size_t(0)
    chk = ctypes.c_uint32(0)
    used = lib().orc_rle_one(p, len(raw), level, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                             ctypes.byref(olen), ctypes.byref(chk))
    return out[:olen.value].tobytes(), int(chk.value), int(used)


def bwt(data, naive=False):
    """-> (bwt_bytes, ptr, has_byte[256] as np.uint8)"""
    n = len(data)
    a, p = _u8(data if n else b"\0")
    out = np.zeros(max(n, 1), dtype=np.uint8)
    hb = np.zeros(256, dtype=np.uint8)
    op = out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    if naive:
        ptr = lib().orc_bwt_naive(p, n, op)
        for c in set(bytes(data)):
            hb[c] = 1
    else:
        ptr = lib().orc_bwt(p, n, op, hb.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    return out[:n].tobytes(), int(ptr), hb


def mtf_and_rle(bwt_bytes, has_byte):
    """-> (syms np.uint16[m], freqs np.uint32[258], num_syms)"""
    n = len(bwt_bytes)
    a, p = _u8(bwt_bytes if n else b"\0")
    hb = np.ascontiguousarray(has_byte, dtype=np.uint8)
    out = np.zeros(n + 2, dtype=np.uint16)
    freqs = np.zeros(258, dtype=np.uint32)
    ns = ctypes.c_uint32(0)
    m = lib().orc_mtf_and_rle(p, n, hb.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                              out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)),
                              freqs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ctypes.byref(ns))
    return out[:m].copy(), freqs, int(ns.value)


def build_table_from_freqs(num_syms, freqs):
    f = np.ascontiguousarray(freqs, dtype=np.uint32)
    out = np.zeros(258, dtype=np.uint8)
    lib().orc_build_table_from_freqs(num_syms, f.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                     out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    return out[:num_syms].copy()


def huffman_block(syms, num_syms, freqs):
    """-> (payload bytes zero padded, nbits, tables np.uint8[ntables,258])"""
    s = np.ascontiguousarray(syms, dtype=np.uint16)
    f = np.ascontiguousarray(freqs, dtype=np.uint32)
    cap = s.size * 3 + 4096
    out = np.zeros(cap, dtype=np.uint8)
    tables = np.zeros(3 * 258, dtype=np.uint8)
    nt = ctypes.c_uint32(0)
    bits = lib().orc_huffman_block(s.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), s.size, num_syms,
                                   f.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                   out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), cap,
                                   tables.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.byref(nt))
    return out[:(bits + 7) // 8].tobytes(), int(bits), tables.reshape(3, 258)[:nt.value].copy()


def encode(data, level=9, want_blocks=False):
    """banzai encode() on an in-memory slice -> stream bytes (and block infos)."""
    n = len(data)
    a, p = _u8(data if n else b"\0")
    cap = n + n // 50 + 4096
    out = np.zeros(cap, dtype=np.uint8)
    olen = ctypes.c_size_t(0)
    maxb = n // (4 * (100000 * level - 1) // 5 - 4) + 8
    infos = (BlockInfo * maxb)()
    nb = ctypes.c_size_t(0)
    used = lib().orc_encode(p, n, level, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), cap,
                            ctypes.byref(olen), infos, maxb, ctypes.byref(nb))
    assert olen.value <= cap, "oracle output overflowed its buffer"
    assert used == n
    stream = out[:olen.value].tobytes()
    if want_blocks:
        return stream, [infos[k] for k in range(nb.value)]
    return stream


class DecodeError(Exception):
    """orc_bz2_decode status: -1 magic, -2 truncated, -3 format, -4 block CRC, -5 stream CRC, -6 capacity,
    -7 memory, -8 trailing bytes"""

    def __init__(self, status):
        super().__init__("bz2 decode status %d" % status)
        self.status = status


def decode(stream, cap=None):
    """In-repo bzip2 decoder (oracle/bz2_decode.c): stream bytes -> original bytes; raises DecodeError."""
    a = np.frombuffer(bytes(stream), dtype=np.uint8) if len(stream) else np.zeros(1, np.uint8)
    if cap is None:
        cap = 64 + len(stream) * 60  # bzip2 cannot beat ~ 1:50 except on runs; grown below when needed
    u8p = ctypes.POINTER(ctypes.c_uint8)
    while True:
        out = np.empty(max(1, cap), dtype=np.uint8)
        n = ctypes.c_size_t(0)
        st = lib().orc_bz2_decode(a.ctypes.data_as(u8p), len(stream), out.ctypes.data_as(u8p), cap, ctypes.byref(n))
        if st == -6 and cap < (1 << 33):
            cap *= 8
            continue
        if st != 0:
            raise DecodeError(st)
        return out[:n.value].tobytes()
