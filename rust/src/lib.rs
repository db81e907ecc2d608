// =-=-= lib.rs =-=-=
// banzai's public surface over libbzhip.so (MI355X).  Signatures are those of the reference
// crate (lib/lib.rs:84-88 and :141-153); every byte is produced by the HIP library.

use std::convert;
use std::ffi::CStr;
use std::fs;
use std::io;
use std::io::{Read, Write};
use std::os::raw::{c_char, c_int};
use std::path;

#[repr(C)]
struct BzhCtx {
    _private: [u8; 0],
}

extern "C" {
    fn bzh_create(ctx: *mut *mut BzhCtx, device: c_int, level: c_int, max_batch: c_int) -> c_int;
    fn bzh_destroy(ctx: *mut BzhCtx);
    fn bzh_strerror(status: c_int) -> *const c_char;
    fn bzh_last_error(ctx: *const BzhCtx) -> *const c_char;
    fn bzh_encode(
        ctx: *mut BzhCtx,
        input: *const u8,
        n: usize,
        out: *mut u8,
        cap: usize,
        out_len: *mut usize,
        consumed: *mut usize,
    ) -> c_int;
}

struct Ctx(*mut BzhCtx);

impl Drop for Ctx {
    fn drop(&mut self) {
        unsafe { bzh_destroy(self.0) }
    }
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

    // Blocks are cut by a sequential rule over the whole input, so the slice is gathered first
    // (a reader that yields everything in one fill_buf is the case the reference handles without
    // its truncation bug).
    let mut raw = vec![];
    reader.read_to_end(&mut raw)?;

    let device: c_int = std::env::var("BZHIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
    let mut handle: *mut BzhCtx = std::ptr::null_mut();
    let status = unsafe { bzh_create(&mut handle, device, level as c_int, 0) };
    if status != 0 {
        return Err(to_io_error(std::ptr::null(), status));
    }
    let ctx = Ctx(handle);

    let cap = raw.len() + raw.len() / 4 + (raw.len() / 70_000 + 2) * 4096 + 65_536;
    let mut out: Vec<u8> = vec![0; cap];
    let (mut out_len, mut consumed) = (0usize, 0usize);
    let status =
        unsafe { bzh_encode(ctx.0, raw.as_ptr(), raw.len(), out.as_mut_ptr(), cap, &mut out_len, &mut consumed) };
    if status != 0 {
        return Err(to_io_error(ctx.0, status));
    }

    writer.write_all(&out[..out_len])?;
    writer.flush()?;
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
    encode(io::BufReader::new(inf), io::BufWriter::new(outf), 9)
}
