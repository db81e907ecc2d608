// =-=-= lib.rs =-=-=
// banzai's public surface over libbzhip.so (MI355X).  Signatures are those of the reference
// crate (lib/lib.rs:84-88 and :141-153); every byte is produced by the HIP library.
// NOTE: the build image has no Rust toolchain, so this crate has never been compiled or run; the same
// calling pattern -- the context pool included -- is executed in C by tests/abi_facade.c (test_rust_facade_twin) and through
// the C ABI by banzai_amd/__init__.py and bnzhip.

use std::collections::HashMap;
use std::convert;
use std::ffi::CStr;
use std::fs;
use std::io;
use std::io::Write;
use std::os::raw::{c_char, c_int};
use std::path;
use std::sync::{Mutex, OnceLock};

#[repr(C)]
struct BzhCtx {
    _private: [u8; 0],
}

extern "C" {
    fn bzh_create(ctx: *mut *mut BzhCtx, device: c_int, level: c_int, max_batch: c_int) -> c_int;
    fn bzh_destroy(ctx: *mut BzhCtx);
    fn bzh_strerror(status: c_int) -> *const c_char;
    fn bzh_last_error(ctx: *const BzhCtx) -> *const c_char;
    fn bzh_stream_begin(ctx: *mut BzhCtx) -> c_int;
    fn bzh_stream_bound(ctx: *const BzhCtx, n: usize) -> usize;
    fn bzh_stream_feed(
        ctx: *mut BzhCtx,
        input: *const u8,
        n: usize,
        eof: c_int,
        out: *mut u8,
        cap: usize,
        out_len: *mut usize,
    ) -> c_int;
    fn bzh_stream_consumed(ctx: *const BzhCtx) -> usize;
}

// A context and the two host buffers an encode works with.  They stay together in the pool: the output buffer is sized by
// bzh_stream_bound (a worst case of a few hundred megabytes for a 100 MB input) and `resize` zero-fills what it adds -- once
// per pooled context, not once per call (per call it cost more than the encode: tests/abi_facade.c measures both).
struct Ctx {
    handle: *mut BzhCtx,
    out: Vec<u8>,
    stage: Vec<u8>,
}

impl Drop for Ctx {
    fn drop(&mut self) {
        unsafe { bzh_destroy(self.handle) }
    }
}

// A context is used by one thread at a time, and may move between threads between calls (include/bzhip.h).
unsafe impl Send for Ctx {}

// Process-wide pool of contexts, keyed by (device, level) -- what banzai_amd/__init__.py keeps in `_contexts`.
// A context owns its workspace arena (45 MB per bzip2 block of a batch: 5.8 GB for a 100 MB input), allocated and
// first-touched on the first encode that needs it; a caller that loops over files through `encode` must not pay that --
// nor `bzh_create` -- per call.  `encode` checks a context OUT of the pool (two threads never share one), and back
// IN when its stream ended cleanly; a context whose stream failed is dropped.  At most POOL_KEEP idle contexts a key.
const POOL_KEEP: usize = 2;
const STAGE: usize = 4 << 20; // small fill_buf slices are coalesced into feeds of this size
static POOL: OnceLock<Mutex<HashMap<(c_int, usize), Vec<Ctx>>>> = OnceLock::new();

fn checkout(device: c_int, level: usize) -> io::Result<Ctx> {
    let pool = POOL.get_or_init(|| Mutex::new(HashMap::new()));
    if let Some(ctx) = pool.lock().unwrap().get_mut(&(device, level)).and_then(|v| v.pop()) {
        return Ok(ctx);
    }
    let mut handle: *mut BzhCtx = std::ptr::null_mut();
    let status = unsafe { bzh_create(&mut handle, device, level as c_int, 0) };
    if status != 0 {
        return Err(to_io_error(std::ptr::null(), status));
    }
    Ok(Ctx { handle, out: Vec::new(), stage: Vec::with_capacity(STAGE) })
}

fn checkin(device: c_int, level: usize, ctx: Ctx) {
    let pool = POOL.get_or_init(|| Mutex::new(HashMap::new()));
    let mut map = pool.lock().unwrap();
    let idle = map.entry((device, level)).or_default();
    if idle.len() < POOL_KEEP {
        idle.push(ctx);
    } // (else: dropped here, bzh_destroy)
}

fn to_io_error(ctx: *const BzhCtx, status: c_int) -> io::Error {
    let text = unsafe {
        let head = CStr::from_ptr(bzh_strerror(status)).to_string_lossy().into_owned();
        if ctx.is_null() {
            head
        } else {
            format!("{}: {}", head, CStr::from_ptr(bzh_last_error(ctx)).to_string_lossy())
        }
    };
    io::Error::new(io::ErrorKind::Other, text)
}

/// bzip2 encode an input stream and write the output to a `BufWriter`
///
/// `level` must be in `1..=9` and describes the block size (`level * 100_000`).
/// Returns the number of input bytes encoded.  Same contract as banzai 0.3.1; the GPU device is
/// taken from `BZHIP_DEVICE` (default 0).
pub fn encode<R, W>(mut reader: R, mut writer: io::BufWriter<W>, level: usize) -> io::Result<usize>
where
    R: io::BufRead,
    W: io::Write,
{
    assert!(1 <= level && level <= 9);

    let device: c_int = std::env::var("BZHIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
    let mut ctx = checkout(device, level)?; // (an error below drops it: only a context whose stream ended cleanly goes back)
    let handle = ctx.handle; // (a copy of the raw pointer: the closure below borrows the context's buffers only)
    let status = unsafe { bzh_stream_begin(handle) };
    if status != 0 {
        return Err(to_io_error(handle, status));
    }

    // Pull from the reader as the reference does (fill_buf / consume, lib/rle.rs:43-91).  A BufReader hands
    // out 8 KiB slices by default (bnz/src/main.rs:263); every bzh_stream_feed call is one blocking H2D copy,
    // so small slices are coalesced into a staging buffer of STAGE bytes and fed from there; slices of STAGE
    // bytes or more go straight through.  The library buffers what it must and returns stream bytes as soon
    // as they are final.
    let out = &mut ctx.out; // (the pooled context's buffers: disjoint fields, borrowed until the loop below ends)
    let stage = &mut ctx.stage;
    stage.clear();
    let mut feed = |chunk: &[u8], eof: bool, writer: &mut io::BufWriter<W>| -> io::Result<()> {
        let cap = unsafe { bzh_stream_bound(handle, chunk.len()) };
        if out.len() < cap {
            out.resize(cap, 0);
        }
        let mut out_len = 0usize;
        let status = unsafe {
            bzh_stream_feed(handle, chunk.as_ptr(), chunk.len(), eof as c_int, out.as_mut_ptr(), out.len(), &mut out_len)
        };
        if status != 0 {
            return Err(to_io_error(handle, status));
        }
        writer.write_all(&out[..out_len])
    };
    loop {
        let len = {
            let buf = reader.fill_buf()?;
            if buf.is_empty() {
                feed(&stage[..], true, &mut writer)?; // end of input: whatever is staged, with the eof mark
                break;
            }
            if stage.is_empty() && buf.len() >= STAGE {
                feed(buf, false, &mut writer)?;
            } else {
                stage.extend_from_slice(buf);
                if stage.len() >= STAGE {
                    feed(&stage[..], false, &mut writer)?;
                    stage.clear();
                }
            }
            buf.len()
        };
        reader.consume(len);
    }
    writer.flush()?;
    let consumed = unsafe { bzh_stream_consumed(handle) };
    checkin(device, level, ctx);
    Ok(consumed)
}

/// bzip2 encode a file and write the output to another file (level 9)
///
/// Returns the number of bytes encoded.
pub fn encode_file<I, O>(in_path: I, out_path: O) -> io::Result<usize>
where
    I: convert::AsRef<path::Path>,
    O: convert::AsRef<path::Path>,
{
    let inf = fs::File::open(in_path.as_ref())?;
    let outf = fs::File::create(out_path.as_ref())?;
    // large reads: each fill_buf slice becomes one H2D copy (see encode)
    encode(io::BufReader::with_capacity(16 << 20, inf), io::BufWriter::new(outf), 9)
}
