"""ctypes binding of libbzhip.so (include/bzhip.h).  No CPU fallback: if the library or a
gfx950 device is missing, calls raise."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# $BZH_LIB selects another build of the same library (A/B timing of kernel variants, scripts/ab.sh)
LIB_PATH = os.environ.get("BZH_LIB") or os.path.join(_HERE, "libbzhip.so")

u8p = ctypes.POINTER(ctypes.c_uint8)
u16p = ctypes.POINTER(ctypes.c_uint16)
u32p = ctypes.POINTER(ctypes.c_uint32)
u64p = ctypes.POINTER(ctypes.c_uint64)
szp = ctypes.POINTER(ctypes.c_size_t)


class BzhError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        super().__init__(f"bzhip status {status}: {detail}")


class Block(ctypes.Structure):
    _fields_ = [("in_off", ctypes.c_uint64), ("in_len", ctypes.c_uint64), ("rle_len", ctypes.c_uint32),
                ("crc", ctypes.c_uint32)]


_BLOCK_DTYPE = np.dtype([("in_off", "<u8"), ("in_len", "<u8"), ("rle_len", "<u4"), ("crc", "<u4")])
assert _BLOCK_DTYPE.itemsize == ctypes.sizeof(Block)


class Stats(ctypes.Structure):
    _fields_ = [(k, ctypes.c_double) for k in
                ("ms_plan", "ms_rle1", "ms_bwt", "ms_mtf", "ms_huff", "ms_pack", "ms_total", "ms_bwt_sort")] + \
               [(k, ctypes.c_uint64) for k in
                ("bwt_sort_launches", "bwt_sort_elems", "raw_bytes", "rle_bytes", "mtf_syms", "out_bits",
                 "bwt_rounds", "bwt_active_sum")] + \
               [("blocks", ctypes.c_uint32), ("pad", ctypes.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "pad"}


class KStat(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("ms", ctypes.c_double), ("launches", ctypes.c_uint64),
                ("alg_bytes", ctypes.c_uint64)]


# name -> (restype, argtypes); also the list the symbol-export test checks against include/bzhip.h
SIGNATURES = {
    "bzh_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "bzh_destroy": (None, [ctypes.c_void_p]),
    "bzh_arch_supported": (ctypes.c_int, [ctypes.c_char_p]),
    "bzh_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "bzh_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "bzh_set_stream": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "bzh_set_profiling": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "bzh_set_lanes": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "bzh_set_mode": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "bzh_get_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Stats)]),
    "bzh_debug_fault": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "bzh_get_kernel_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(KStat), ctypes.c_size_t, szp]),
    "bzh_encode": (ctypes.c_int, [ctypes.c_void_p, u8p, ctypes.c_size_t, u8p, ctypes.c_size_t, szp, szp]),
    "bzh_encode_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                         ctypes.c_size_t, szp, szp]),
    "bzh_stream_begin": (ctypes.c_int, [ctypes.c_void_p]),
    "bzh_stream_feed": (ctypes.c_int, [ctypes.c_void_p, u8p, ctypes.c_size_t, ctypes.c_int, u8p, ctypes.c_size_t, szp]),
    "bzh_stream_bound": (ctypes.c_size_t, [ctypes.c_void_p, ctypes.c_size_t]),
    "bzh_stream_set_chunk": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    "bzh_stream_consumed": (ctypes.c_size_t, [ctypes.c_void_p]),
    "bzh_plan_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, szp]),
    "bzh_plan_tables_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "bzh_plan_split_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, szp]),
    "bzh_plan_blocks": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Block), ctypes.c_size_t]),
    "bzh_plan_device_nocrc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, szp]),
    "bzh_plan_crc_range": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t]),
    "bzh_plan_open": (ctypes.c_int, [ctypes.c_void_p, u8p, ctypes.c_size_t]),
    "bzh_encode_range_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p,
                                               ctypes.c_size_t, u64p]),
    "bzh_assemble_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), u64p, ctypes.c_size_t,
                                           u32p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, szp]),
    "bzh_create_multi": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int]),
    "bzh_destroy_multi": (None, [ctypes.c_void_p]),
    "bzh_multi_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "bzh_multi_device_count": (ctypes.c_int, [ctypes.c_void_p]),
    "bzh_multi_encode": (ctypes.c_int, [ctypes.c_void_p, u8p, ctypes.c_size_t, u8p, ctypes.c_size_t, szp, szp]),
    "bzh_multi_load": (ctypes.c_int, [ctypes.c_void_p, u8p, ctypes.c_size_t]),
    "bzh_multi_run": (ctypes.c_int, [ctypes.c_void_p, szp]),
    "bzh_multi_fetch": (ctypes.c_int, [ctypes.c_void_p, u8p, ctypes.c_size_t]),
    "bzh_multi_output_device": (ctypes.c_void_p, [ctypes.c_void_p]),
    "bzh_multi_times": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_size_t]),
    "bzh_multi_debug_slab": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    "bzh_rle1_split": (ctypes.c_int, [ctypes.c_void_p, u8p, ctypes.c_size_t, ctypes.POINTER(Block), ctypes.c_size_t,
                                      szp, u8p, ctypes.c_size_t]),
    "bzh_crc32": (ctypes.c_int, [ctypes.c_void_p, u8p, ctypes.c_size_t, u32p]),
    "bzh_bwt_batch": (ctypes.c_int, [ctypes.c_void_p, u8p, u64p, u32p, ctypes.c_size_t, u8p, u32p, u8p]),
    "bzh_unbwt_batch": (ctypes.c_int, [ctypes.c_void_p, u8p, u64p, u32p, u32p, ctypes.c_size_t, u8p]),
    "bzh_bwt_roundtrip_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, u64p]),
    "bzh_bwt": (ctypes.c_int, [ctypes.c_void_p, u8p, ctypes.c_size_t, u8p, u32p, u8p]),
    "bzh_mtf": (ctypes.c_int, [ctypes.c_void_p, u8p, ctypes.c_size_t, u8p, u16p, szp, u32p, u32p]),
    "bzh_huffman": (ctypes.c_int, [ctypes.c_void_p, u16p, ctypes.c_size_t, ctypes.c_uint32, u32p, u8p,
                                   ctypes.c_size_t, u64p, u8p, u32p]),
}


def build(force=False):
    """Compile libbzhip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "bzhip.h"))
    newest = max(os.path.getmtime(s) for s in srcs)
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < newest:
        subprocess.check_call(["make", "-C", src_dir, "-s", "-j4"])
    return LIB_PATH


_lib = None
MISSING = []


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BzhError(-3, f"{LIB_PATH} is missing: build it with banzai_amd._native.build() "
                               "(there is no CPU fallback)")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                MISSING.append(name)  # the export test requires this list to stay empty
                continue
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def ptr(a, t=u8p):
    return a.ctypes.data_as(t)


class Context:
    """One context per GPU (bzh_create / bzh_destroy)."""

    def __init__(self, device=0, level=9, max_batch=0):
        self._h = ctypes.c_void_p()
        self.level = level
        self._destroy = lib().bzh_destroy  # kept so that close() still works during interpreter shutdown
        st = lib().bzh_create(ctypes.byref(self._h), device, level, max_batch)
        if st != 0:
            self._h = None
            raise BzhError(st, lib().bzh_strerror(st).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def check(self, st):
        if st != 0:
            raise BzhError(st, lib().bzh_strerror(st).decode() + ": " + lib().bzh_last_error(self._h).decode())

    @property
    def handle(self):
        return self._h

    def set_stream(self, stream_ptr):
        self.check(lib().bzh_set_stream(self._h, ctypes.c_void_p(stream_ptr)))

    def set_profiling(self, on=True):
        self.check(lib().bzh_set_profiling(self._h, 1 if on else 0))

    def set_mode(self, fixed):
        """False: the reference's Huffman behaviour (default, bit-identical); True: the opt-in "fixed" mode"""
        self.check(lib().bzh_set_mode(self._h, 1 if fixed else 0))

    def set_lanes(self, lanes):
        self.check(lib().bzh_set_lanes(self._h, lanes))

    def stats(self):
        s = Stats()
        self.check(lib().bzh_get_stats(self._h, ctypes.byref(s)))
        return s.as_dict()

    def debug_fault(self, kind):
        self.check(lib().bzh_debug_fault(self._h, kind))

    def kernel_stats(self):
        """Per-kernel-class (name, ms, launches, algorithmic bytes) of the last call made with profiling on."""
        arr = (KStat * 64)()
        cnt = ctypes.c_size_t(0)
        self.check(lib().bzh_get_kernel_stats(self._h, arr, 64, ctypes.byref(cnt)))
        return [{"name": arr[k].name.decode(), "ms": arr[k].ms, "launches": int(arr[k].launches),
                 "alg_bytes": int(arr[k].alg_bytes)} for k in range(cnt.value)]

    # ---- stage seams (host numpy in / out) ----
    def bwt(self, data):
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        n = a.size
        src = np.ascontiguousarray(a) if n else np.zeros(1, np.uint8)
        out = np.zeros(max(n, 1), dtype=np.uint8)
        p = ctypes.c_uint32(0)
        hb = np.zeros(256, dtype=np.uint8)
        self.check(lib().bzh_bwt(self._h, ptr(src), n, ptr(out), ctypes.byref(p), ptr(hb)))
        return out[:n].tobytes(), int(p.value), hb

    def bwt_batch(self, blocks):
        lens = np.array([len(b) for b in blocks], dtype=np.uint32)
        offs = np.zeros(len(blocks), dtype=np.uint64)
        offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
        cat = np.frombuffer(b"".join(bytes(b) for b in blocks), dtype=np.uint8).copy()
        out = np.zeros_like(cat)
        ptrs = np.zeros(len(blocks), dtype=np.uint32)
        hb = np.zeros(len(blocks) * 256, dtype=np.uint8)
        self.check(lib().bzh_bwt_batch(self._h, ptr(cat), ptr(offs, u64p), ptr(lens, u32p), len(blocks), ptr(out),
                                       ptr(ptrs, u32p), ptr(hb)))
        res = []
        for k in range(len(blocks)):
            o = int(offs[k])
            res.append((out[o:o + int(lens[k])].tobytes(), int(ptrs[k]), hb[k * 256:(k + 1) * 256].copy()))
        return res

    def unbwt_batch(self, blocks):
        """blocks: [(bwt bytes, ptr)] -> [original bytes], computed on the GPU (bzh_unbwt_batch)"""
        lens = np.array([len(b) for b, _ in blocks], dtype=np.uint32)
        ptrs = np.array([p for _, p in blocks], dtype=np.uint32)
        offs = np.zeros(len(blocks), dtype=np.uint64)
        offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
        cat = np.frombuffer(b"".join(bytes(b) for b, _ in blocks), dtype=np.uint8).copy()
        out = np.zeros_like(cat)
        self.check(lib().bzh_unbwt_batch(self._h, ptr(cat), ptr(offs, u64p), ptr(lens, u32p), ptr(ptrs, u32p),
                                         len(blocks), ptr(out)))
        return [out[int(offs[k]):int(offs[k]) + int(lens[k])].tobytes() for k in range(len(blocks))]

    def bwt_roundtrip_device(self, b0, b1):
        """forward + inverse BWT of plan blocks [b0, b1) on the device -> number of mismatching bytes"""
        bad = ctypes.c_uint64(0)
        self.check(lib().bzh_bwt_roundtrip_device(self._h, b0, b1, ctypes.byref(bad)))
        return int(bad.value)

    def mtf(self, bwt_bytes, has_byte):
        a = np.frombuffer(bytes(bwt_bytes), dtype=np.uint8).copy()
        n = a.size
        hb = np.ascontiguousarray(has_byte, dtype=np.uint8)
        syms = np.zeros(n + 2, dtype=np.uint16)
        freqs = np.zeros(258, dtype=np.uint32)
        m = ctypes.c_size_t(0)
        ns = ctypes.c_uint32(0)
        self.check(lib().bzh_mtf(self._h, ptr(a), n, ptr(hb), ptr(syms, u16p), ctypes.byref(m), ptr(freqs, u32p),
                                 ctypes.byref(ns)))
        return syms[:m.value].copy(), freqs, int(ns.value)

    def huffman(self, syms, num_syms, freqs):
        s = np.ascontiguousarray(syms, dtype=np.uint16)
        f = np.ascontiguousarray(freqs, dtype=np.uint32)
        cap = s.size * 3 + 8192
        out = np.zeros(cap, dtype=np.uint8)
        lens = np.zeros(3 * 258, dtype=np.uint8)
        nb = ctypes.c_uint64(0)
        nt = ctypes.c_uint32(0)
        self.check(lib().bzh_huffman(self._h, ptr(s, u16p), s.size, num_syms, ptr(f, u32p), ptr(out), cap,
                                     ctypes.byref(nb), ptr(lens), ctypes.byref(nt)))
        return out[:(nb.value + 7) // 8].tobytes(), int(nb.value), lens.reshape(3, 258)[:nt.value].copy()

    def crc32(self, data):
        a = np.frombuffer(bytes(data), dtype=np.uint8).copy() if len(data) else np.zeros(1, np.uint8)
        c = ctypes.c_uint32(0)
        self.check(lib().bzh_crc32(self._h, ptr(a), len(data), ctypes.byref(c)))
        return int(c.value)

    def rle1_split(self, data, want_bytes=True):
        """-> ([(in_off, in_len, rle_len, crc)], [rle bytes per block])"""
        n = len(data)
        a = np.frombuffer(bytes(data), dtype=np.uint8).copy() if n else np.zeros(1, np.uint8)
        maxb = n // (4 * (100000 * self.level - 1) // 5 - 4) + 8
        blocks = (Block * maxb)()
        nb = ctypes.c_size_t(0)
        cap = n + n // 4 + 64
        out = np.zeros(cap if want_bytes else 1, dtype=np.uint8)
        self.check(lib().bzh_rle1_split(self._h, ptr(a), n, blocks, maxb, ctypes.byref(nb),
                                        ptr(out) if want_bytes else None, cap))
        infos = [(int(blocks[k].in_off), int(blocks[k].in_len), int(blocks[k].rle_len), int(blocks[k].crc))
                 for k in range(nb.value)]
        chunks = []
        if want_bytes:
            pos = 0
            for (_, _, rl, _) in infos:
                chunks.append(out[pos:pos + rl].tobytes())
                pos += rl
        return infos, chunks

    def encode(self, data):
        """bzh_encode: complete .bz2 stream of `data` (host buffers)."""
        n = len(data)
        a = np.frombuffer(bytes(data), dtype=np.uint8).copy() if n else np.zeros(1, np.uint8)
        cap = n + n // 4 + 65536 + (n // 70000 + 2) * 4096
        out = np.zeros(cap, dtype=np.uint8)
        olen = ctypes.c_size_t(0)
        used = ctypes.c_size_t(0)
        self.check(lib().bzh_encode(self._h, ptr(a), n, ptr(out), cap, ctypes.byref(olen), ctypes.byref(used)))
        assert used.value == n
        return out[:olen.value].tobytes()

    def encode_host_ptr(self, in_ptr, n, out_ptr, cap):
        """bzh_encode on raw host addresses (e.g. pinned buffers): H2D + encode + D2H.  -> stream length."""
        olen = ctypes.c_size_t(0)
        used = ctypes.c_size_t(0)
        self.check(lib().bzh_encode(self._h, ctypes.cast(in_ptr, u8p), n, ctypes.cast(out_ptr, u8p), cap,
                                    ctypes.byref(olen), ctypes.byref(used)))
        return int(olen.value)

    def encode_device(self, d_in, n, d_out, cap):
        """Device-resident encode; d_in/d_out are integer device addresses.  -> stream length."""
        olen = ctypes.c_size_t(0)
        used = ctypes.c_size_t(0)
        self.check(lib().bzh_encode_device(self._h, ctypes.c_void_p(d_in), n, ctypes.c_void_p(d_out), cap,
                                           ctypes.byref(olen), ctypes.byref(used)))
        return int(olen.value)

    def plan_device(self, d_in, n, crc=True):
        """-> [(in_off, in_len, rle_len, crc)]; crc=False leaves the CRCs (0 here) to encode_range_device /
        plan_crc_range (the sharded path: a rank only needs the CRCs of the blocks it encodes)"""
        nb = ctypes.c_size_t(0)
        fn = lib().bzh_plan_device if crc else lib().bzh_plan_device_nocrc
        self.check(fn(self._h, ctypes.c_void_p(d_in), n, ctypes.byref(nb)))
        self._nblocks = nb.value
        return self.plan_blocks()

    def plan_device_only(self, d_in, n, crc=True):
        """bzh_plan_device[_nocrc] without turning the block table into Python objects -> number of blocks"""
        nb = ctypes.c_size_t(0)
        fn = lib().bzh_plan_device if crc else lib().bzh_plan_device_nocrc
        self.check(fn(self._h, ctypes.c_void_p(d_in), n, ctypes.byref(nb)))
        self._nblocks = nb.value
        return int(nb.value)

    def plan_tables_device(self, d_in, n):
        """bzh_plan_tables_device: the split's run tables over d_in[0..n) (queued, no wait)"""
        self.check(lib().bzh_plan_tables_device(self._h, ctypes.c_void_p(d_in), n))

    def plan_split_device(self, start, stop=None, crc=False):
        """bzh_plan_split_device: cut blocks from offset `start` of the buffer of plan_tables_device until one starts at
        or after `stop` (None: to the end) -> number of blocks"""
        nb = ctypes.c_size_t(0)
        stop = ctypes.c_size_t(-1).value if stop is None else stop
        self.check(lib().bzh_plan_split_device(self._h, start, stop, 1 if crc else 0, ctypes.byref(nb)))
        self._nblocks = nb.value
        return int(nb.value)

    def plan_blocks_np(self):
        """the plan as one structured numpy array (fields in_off, in_len, rle_len, crc): no per-block Python objects"""
        n = getattr(self, "_nblocks", 0)
        blocks = (Block * max(1, n))()
        self.check(lib().bzh_plan_blocks(self._h, blocks, max(1, n)))
        return np.frombuffer(blocks, dtype=_BLOCK_DTYPE, count=n).copy()

    def plan_open_np(self):
        n = getattr(self, "_nblocks", 0)
        flags = np.zeros(max(1, n), dtype=np.uint8)
        self.check(lib().bzh_plan_open(self._h, ptr(flags), max(1, n)))
        return flags[:n].astype(bool)

    def plan_blocks(self):
        n = getattr(self, "_nblocks", 0)
        blocks = (Block * max(1, n))()
        self.check(lib().bzh_plan_blocks(self._h, blocks, max(1, n)))
        # one numpy view instead of 4 ctypes field reads per block (≈ 1 ms per 1000 blocks otherwise)
        arr = np.frombuffer(blocks, dtype=_BLOCK_DTYPE, count=n)
        return list(zip(arr["in_off"].tolist(), arr["in_len"].tolist(), arr["rle_len"].tolist(), arr["crc"].tolist()))

    def plan_open(self):
        """per block of the last plan: True if its cut could still move were the input longer"""
        n = getattr(self, "_nblocks", 0)
        flags = np.zeros(max(1, n), dtype=np.uint8)
        self.check(lib().bzh_plan_open(self._h, ptr(flags), max(1, n)))
        return [bool(x) for x in flags[:n]]

    def plan_crc_range(self, b0, b1):
        """CRCs of plan blocks [b0, b1) (computed now unless already known) -> [crc]"""
        self.check(lib().bzh_plan_crc_range(self._h, b0, b1))
        return [b[3] for b in self.plan_blocks()[b0:b1]]

    def encode_range_device(self, b0, b1, d_out, cap):
        nbits = ctypes.c_uint64(0)
        self.check(lib().bzh_encode_range_device(self._h, b0, b1, ctypes.c_void_p(d_out), cap, ctypes.byref(nbits)))
        return int(nbits.value)

    def assemble_device(self, segs, crcs, d_out, cap):
        """segs: [(device address, nbits)], crcs: block CRCs in block order -> stream length."""
        nseg = len(segs)
        ptrs = (ctypes.c_void_p * max(1, nseg))(*[ctypes.c_void_p(p) for p, _ in segs])
        bits = np.array([b for _, b in segs] or [0], dtype=np.uint64)
        c = np.array(list(crcs) or [0], dtype=np.uint32)
        olen = ctypes.c_size_t(0)
        self.check(lib().bzh_assemble_device(self._h, ptrs, ptr(bits, u64p), nseg, ptr(c, u32p), len(crcs),
                                             ctypes.c_void_p(d_out), cap, ctypes.byref(olen)))
        return int(olen.value)

    # ---- streaming (bzh_stream_*) ----
    def stream_begin(self, chunk_bytes=None):
        if chunk_bytes is not None:
            self.check(lib().bzh_stream_set_chunk(self._h, chunk_bytes))
        self.check(lib().bzh_stream_begin(self._h))
        # (the output buffer of the previous stream is kept: a fresh numpy array costs a page fault per 4 KiB the library
        # writes -- several milliseconds per stream for the ~30 MB a 100 MB input produces)
        if not hasattr(self, "_sbuf"):
            self._sbuf = None

    def stream_feed(self, data, eof=False):
        """-> stream bytes that became final with this feed"""
        n = len(data)
        a = np.frombuffer(data, dtype=np.uint8) if n else np.zeros(1, np.uint8)  # zero-copy view of any buffer
        need = int(lib().bzh_stream_bound(self._h, n))
        if self._sbuf is None or self._sbuf.size < need:
            self._sbuf = np.empty(need, dtype=np.uint8)
        got = ctypes.c_size_t(0)
        self.check(lib().bzh_stream_feed(self._h, ptr(np.ascontiguousarray(a)), n, 1 if eof else 0, ptr(self._sbuf),
                                         self._sbuf.size, ctypes.byref(got)))
        return self._sbuf[:got.value].tobytes()

    def stream_feed_view(self, data, eof=False):
        """as stream_feed, but returns a memoryview of the context's output buffer (valid until the next feed)"""
        n = len(data)
        a = np.frombuffer(data, dtype=np.uint8) if n else np.zeros(1, np.uint8)
        need = int(lib().bzh_stream_bound(self._h, n))
        if self._sbuf is None or self._sbuf.size < need:
            self._sbuf = np.empty(need, dtype=np.uint8)
        got = ctypes.c_size_t(0)
        self.check(lib().bzh_stream_feed(self._h, ptr(a), n, 1 if eof else 0, ptr(self._sbuf), self._sbuf.size,
                                         ctypes.byref(got)))
        return memoryview(self._sbuf)[:got.value]

    def stream_consumed(self):
        return int(lib().bzh_stream_consumed(self._h))


class MultiContext:
    """Several GPUs behind one handle (bzh_create_multi): one host thread and one context per listed device inside the
    library, block ranges chained by start offset, the bit strings assembled on devices[0].  A device may be listed more
    than once (one context per entry)."""

    def __init__(self, devices, level=9):
        self.devices = [int(d) for d in devices]
        self.level = level
        self._h = ctypes.c_void_p()
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        st = lib().bzh_create_multi(ctypes.byref(self._h), arr, len(self.devices), level)
        if st != 0:
            raise BzhError(st, lib().bzh_strerror(st).decode())

    def close(self):
        if self._h:
            lib().bzh_destroy_multi(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def check(self, st):
        if st != 0:
            raise BzhError(st, lib().bzh_strerror(st).decode() + ": " + lib().bzh_multi_last_error(self._h).decode())

    @staticmethod
    def _in(data):
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        return a if a.size else np.zeros(1, dtype=np.uint8), int(a.size)

    def encode(self, data):
        """bytes-like -> the whole .bz2 stream (bytes)"""
        a, n = self._in(data)
        cap = n + n // 4 + (n // 70000 + 4) * 4096 + 65536
        out = np.empty(cap, dtype=np.uint8)
        got, used = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self.check(lib().bzh_multi_encode(self._h, ptr(a), n, ptr(out), cap, ctypes.byref(got), ctypes.byref(used)))
        return out[:got.value].tobytes()

    def load(self, data):
        a, n = self._in(data)
        self.check(lib().bzh_multi_load(self._h, ptr(a), n))

    def run(self):
        got = ctypes.c_size_t(0)
        self.check(lib().bzh_multi_run(self._h, ctypes.byref(got)))
        return got.value

    def fetch(self, n):
        out = np.empty(max(1, n), dtype=np.uint8)
        self.check(lib().bzh_multi_fetch(self._h, ptr(out), out.size))
        return out[:n].tobytes()

    def debug_slab(self, nbytes):
        self.check(lib().bzh_multi_debug_slab(self._h, nbytes))

    def times(self):
        w = len(self.devices)
        buf = (ctypes.c_double * (5 * w))()
        self.check(lib().bzh_multi_times(self._h, buf, w))
        keys = ("ms_load", "ms_wait", "ms_plan", "ms_encode", "ms_copy")
        return [dict(zip(keys, (round(buf[5 * r + k], 3) for k in range(5)))) for r in range(w)]
