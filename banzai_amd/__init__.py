"""banzai_amd -- MI355X-native bzip2 block encoder with banzai's API and banzai's bits.

Mirrors the public surface of jgbyrne/banzai v0.3.1 (reference lib/lib.rs:84-153):

    encode(reader, writer, level) -> bytes consumed      (lib/lib.rs:84-132)
    encode_file(in_path, out_path) -> bytes consumed     (lib/lib.rs:141-153, level 9)

Everything is computed by hand-written HIP kernels behind the C ABI in include/bzhip.h
(libbzhip.so); there is no CPU path.  `reader` is any object with .read(), `writer` any object
with .write() (the Rust signature takes BufRead / BufWriter<W>).
"""
import io

from . import _native

__all__ = ["encode", "encode_file", "Context", "MultiContext", "BzhError"]

Context = _native.Context
MultiContext = _native.MultiContext
BzhError = _native.BzhError

_ctx_cache = {}
_multi_cache = {}
READ_CHUNK = 16 << 20


def _ctx(level, device=0):
    key = (device, level)
    if key not in _ctx_cache:
        _ctx_cache[key] = _native.Context(device, level, 0)
    return _ctx_cache[key]


def _copying_sink(writer):
    """True for the writers that are known to copy what write() is handed before it returns (in-memory and real files):
    only they may be given a VIEW of the context's output buffer, which the next feed overwrites.  Any other writer -- a
    list's append, a queue, a transport -- may keep the object, so it gets its own bytes."""
    return isinstance(writer, (io.BytesIO, io.BufferedWriter, io.BufferedRandom, io.FileIO))


def _device_list(devices):
    """`devices=` of encode, or $BZHIP_DEVICES ("0,1,2,3"): the GPUs of one node an encode is spread over."""
    if devices is None:
        import os
        env = os.environ.get("BZHIP_DEVICES", "").strip()
        devices = [int(x) for x in env.split(",") if x.strip() != ""] if env else None
    return [int(d) for d in devices] if devices else None


def encode(reader, writer, level, device=0, devices=None):
    """bzip2-encode everything `reader` yields and write the stream to `writer`.

    Same contract as banzai::encode: `level` in 1..=9 is the block size in 100 kB units
    (anything else raises, the reference asserts at lib/lib.rs:89); returns the number of input
    bytes encoded; I/O errors of reader/writer propagate.

    `devices` (or $BZHIP_DEVICES): a list of HIP devices of this node -- the blocks are then cut and encoded on all of
    them (bzh_create_multi: one host thread and context per device inside the library, the stream assembled on the
    first) and the stream is the one a single device writes, bit for bit.  That path reads the whole input first."""
    if isinstance(level, bool) or not isinstance(level, int) or not 1 <= level <= 9:
        raise ValueError("level must be in 1..=9")
    devs = _device_list(devices)
    if devs and len(devs) > 1:
        data = reader.getvalue()[reader.tell():] if isinstance(reader, io.BytesIO) else reader.read()
        if not isinstance(data, (bytes, bytearray, memoryview)):
            raise TypeError("reader.read() must return bytes")
        key = (tuple(devs), level)
        if key not in _multi_cache:
            _multi_cache[key] = _native.MultiContext(devs, level)
        writer.write(_multi_cache[key].encode(data))
        if isinstance(reader, io.BytesIO):
            reader.seek(0, io.SEEK_END)
        if hasattr(writer, "flush"):
            writer.flush()
        return len(data)
    if devs:
        device = devs[0]
    ctx = _ctx(level, device)
    put = writer.write if _copying_sink(writer) else (lambda view: writer.write(bytes(view)))
    # incremental ingestion (the reference pulls from fill_buf as it goes, lib/rle.rs:30-92): input is
    # handed to the GPU in chunks, finished stream bytes are written as soon as they are final
    ctx.stream_begin()
    # An in-memory reader (io.BytesIO) lends its buffer, as BufRead::fill_buf lends the reference a slice of the
    # reader's own buffer (lib/rle.rs:30-92): the chunks go to the GPU from where they lie, no copy on this side.
    if isinstance(reader, io.BytesIO):
        # (getvalue() hands out the bytes object BytesIO holds -- no copy for a reader made from bytes; getbuffer()
        # would first un-share it, i.e. copy all of it)
        view = memoryview(reader.getvalue())
        pos, end = reader.tell(), len(view)
        while True:
            k = min(READ_CHUNK, max(0, end - pos))
            out = ctx.stream_feed_view(view[pos:pos + k], k == 0)
            pos += k
            if len(out):
                put(out)
            if k == 0:
                break
        reader.seek(pos)
        if hasattr(writer, "flush"):
            writer.flush()
        return ctx.stream_consumed()
    # one reusable buffer: a reader with readinto() fills it in place (no bytes object per chunk); finished stream
    # bytes go to a copying sink (BytesIO, a real file) as a view of the context's output buffer, to any other writer as bytes
    buf = bytearray(READ_CHUNK) if hasattr(reader, "readinto") else None
    while True:
        chunk = None
        if buf is not None:
            try:
                got = reader.readinto(buf)
            except (NotImplementedError, io.UnsupportedOperation):  # e.g. a RawIOBase subclass that only defines read()
                buf = None
            else:
                if got is None:
                    raise TypeError("reader.readinto() must return a byte count (blocking reader expected)")
                chunk = memoryview(buf)[:got]
        if chunk is None:
            chunk = reader.read(READ_CHUNK)
            if not isinstance(chunk, (bytes, bytearray, memoryview)):
                raise TypeError("reader.read() must return bytes")
        eof = len(chunk) == 0
        out = ctx.stream_feed_view(chunk, eof)
        if len(out):
            put(out)
        if eof:
            break
    if hasattr(writer, "flush"):
        writer.flush()
    return ctx.stream_consumed()


def encode_file(in_path, out_path, device=0, devices=None):
    """bzip2-encode a file into another file at level 9 (banzai::encode_file)."""
    with open(in_path, "rb") as inf, open(out_path, "wb") as outf:
        return encode(inf, outf, 9, device, devices)
