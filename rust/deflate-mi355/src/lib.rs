//! deflate-rs 1.0.0's public surface (src/lib.rs:137-286, src/writer.rs:89-467,
//! src/compression_options.rs, src/lz77.rs:27-37) forwarding to the C ABI of include/mi355_deflate.h.
//! Every item names the reference item it mirrors.  The encoders behave like the reference's: bytes
//! reach the inner writer at flush() / finish(), a writer that takes fewer bytes than offered is retried
//! (src/compress.rs:96-124, 280-299), Drop finishes the stream (src/writer.rs:139-152).
use std::io::{self, Write};
use std::os::raw::c_int;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct Mi355Opts {
    // include/mi355_deflate.h: mi355_deflate_opts
    max_hash_checks: u16,
    lazy_if_less_than: u16,
    matching_type: u8, // 0 Greedy, 1 Lazy
    wrapper: u8,       // 0 raw, 1 zlib, 2 gzip
    compat: u8,
    flush: u8, // 0 Finish, 1 Sync
}
#[repr(C)]
pub struct Ctx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct Stream {
    _p: [u8; 0],
}
#[repr(C)]
pub struct Multi {
    _p: [u8; 0],
}

extern "C" {
    fn mi355_deflate_bound(in_len: usize) -> usize;
    fn mi355_deflate_bound_ex(in_len: usize, wrapper: c_int, hdr_len: usize, n_flush: usize) -> usize;
    fn mi355_deflate_encode(ctx: *mut Ctx, input: *const u8, in_len: usize, opts: *const Mi355Opts, out: *mut u8,
                            out_cap: usize, out_len: *mut usize) -> c_int;
    fn mi355_deflate_encode_gzip(ctx: *mut Ctx, input: *const u8, in_len: usize, opts: *const Mi355Opts, hdr: *const u8,
                                 hdr_len: usize, out: *mut u8, out_cap: usize, out_len: *mut usize) -> c_int;
    fn mi355_deflate_stream_new(ctx: *mut Ctx, opts: *const Mi355Opts, out: *mut *mut Stream) -> c_int;
    fn mi355_deflate_stream_write(s: *mut Stream, data: *const u8, n: usize) -> c_int;
    fn mi355_deflate_stream_flush(s: *mut Stream) -> c_int;
    fn mi355_deflate_stream_finish(s: *mut Stream) -> c_int;
    fn mi355_deflate_stream_output(s: *mut Stream, data: *mut *const u8, n: *mut usize) -> c_int;
    fn mi355_deflate_stream_take_output(s: *mut Stream, dst: *mut u8, cap: usize, n: *mut usize) -> c_int;
    fn mi355_deflate_stream_checksum(s: *mut Stream, sum: *mut u32) -> c_int;
    fn mi355_deflate_stream_reset(s: *mut Stream, data: *mut *const u8, n: *mut usize) -> c_int;
    fn mi355_deflate_stream_gzip_header(s: *mut Stream, hdr: *const u8, n: usize) -> c_int;
    fn mi355_deflate_stream_free(s: *mut Stream);
    // one input over all GPUs of the node in one call (include/mi355_deflate.h, "ONE input over the GPUs of a node")
    fn mi355_multi_create(devices: *const c_int, n_devices: c_int, out: *mut *mut Multi) -> c_int;
    fn mi355_deflate_encode_multi(m: *mut Multi, input: *const u8, in_len: usize, opts: *const Mi355Opts, gz_hdr: *const u8,
                                  gz_len: usize, out: *mut u8, out_cap: usize, out_len: *mut usize) -> c_int;
    fn mi355_device_count() -> c_int; // (the shim links libmi355deflate.so only: no HIP symbol is named from Rust)
}

/// Inputs of at least this many bytes are cut over all GPUs of the node by the one-shot functions (below it one GPU is
/// as fast: a rank should have tens of megabytes).  MI355_DEFLATE_GPUS overrides the device count (1 = never shard).
pub const MULTI_GPU_FROM: usize = 256 << 20;

struct MultiHandle(*mut Multi);
unsafe impl Send for MultiHandle {}
/// the node's handle: one context and one host thread per device, made on first use and kept for the process
fn multi() -> Option<&'static std::sync::Mutex<MultiHandle>> {
    use std::sync::{Mutex, OnceLock};
    static M: OnceLock<Option<Mutex<MultiHandle>>> = OnceLock::new();
    M.get_or_init(|| unsafe {
        let mut n: c_int = mi355_device_count();
        if let Some(v) = std::env::var("MI355_DEFLATE_GPUS").ok().and_then(|v| v.parse::<c_int>().ok()) {
            n = n.min(v);
        }
        if n < 2 {
            return None;
        }
        let devs: Vec<c_int> = (0..n).collect();
        let mut h: *mut Multi = std::ptr::null_mut();
        if mi355_multi_create(devs.as_ptr(), n, &mut h) != 0 {
            return None;
        }
        Some(Mutex::new(MultiHandle(h)))
    })
    .as_ref()
}

/// src/lz77.rs:27-37
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum MatchingType {
    Greedy,
    Lazy,
}
/// src/compression_options.rs:31-42
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Compression {
    Fast,
    Default,
    Best,
}
/// src/compression_options.rs:78-120
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct CompressionOptions {
    pub max_hash_checks: u16,
    pub lazy_if_less_than: u16,
    pub matching_type: MatchingType,
}
impl CompressionOptions {
    /// :126-133
    pub const fn high() -> Self {
        Self { max_hash_checks: 1768, lazy_if_less_than: 128, matching_type: MatchingType::Lazy }
    }
    /// :141-148
    pub const fn fast() -> Self {
        Self { max_hash_checks: 1, lazy_if_less_than: 0, matching_type: MatchingType::Greedy }
    }
    /// :155-162
    pub const fn huffman_only() -> Self {
        Self { max_hash_checks: 0, lazy_if_less_than: 0, matching_type: MatchingType::Greedy }
    }
    /// :171-178
    pub const fn rle() -> Self {
        Self { max_hash_checks: 0, lazy_if_less_than: 0, matching_type: MatchingType::Lazy }
    }
}
impl Default for CompressionOptions {
    /// :67-72
    fn default() -> Self {
        Self { max_hash_checks: 128, lazy_if_less_than: 32, matching_type: MatchingType::Lazy }
    }
}
impl From<Compression> for CompressionOptions {
    /// :188-196
    fn from(c: Compression) -> Self {
        match c {
            Compression::Fast => Self::fast(),
            Compression::Default => Self::default(),
            Compression::Best => Self::high(),
        }
    }
}
fn c_opts(o: CompressionOptions, wrapper: u8) -> Mi355Opts {
    Mi355Opts {
        max_hash_checks: o.max_hash_checks,
        lazy_if_less_than: o.lazy_if_less_than,
        matching_type: (o.matching_type == MatchingType::Lazy) as u8,
        wrapper,
        compat: 0,
        flush: 0,
    }
}

fn one_shot(input: &[u8], o: Mi355Opts) -> Vec<u8> {
    unsafe {
        let cap = mi355_deflate_bound(input.len());
        let mut out = Vec::<u8>::with_capacity(cap);
        let mut n = 0usize;
        // a large input goes over all GPUs of the node: the same bytes (stream-exact), one call
        if input.len() >= MULTI_GPU_FROM {
            if let Some(m) = multi() {
                let g = m.lock().unwrap();
                let rc = mi355_deflate_encode_multi(g.0, input.as_ptr(), input.len(), &o, std::ptr::null(), 0, out.as_mut_ptr(), cap, &mut n);
                if rc == 0 {
                    out.set_len(n);
                    return out;
                }
                // (a device that ran out of memory, a peer that cannot be reached ...: the single-device call takes any length)
            }
        }
        let rc = mi355_deflate_encode(std::ptr::null_mut(), input.as_ptr(), input.len(), &o, out.as_mut_ptr(), cap, &mut n);
        assert!(rc == 0, "mi355_deflate_encode failed: {}", rc); // lib.rs:145 expect("Write error!")
        out.set_len(n);
        out
    }
}
/// src/lib.rs:137-147
pub fn deflate_bytes_conf<O: Into<CompressionOptions>>(input: &[u8], options: O) -> Vec<u8> {
    one_shot(input, c_opts(options.into(), 0))
}
/// src/lib.rs:163-165
pub fn deflate_bytes(input: &[u8]) -> Vec<u8> {
    deflate_bytes_conf(input, Compression::Default)
}
/// src/lib.rs:182-198
pub fn deflate_bytes_zlib_conf<O: Into<CompressionOptions>>(input: &[u8], options: O) -> Vec<u8> {
    one_shot(input, c_opts(options.into(), 1))
}
/// src/lib.rs:216-218
pub fn deflate_bytes_zlib(input: &[u8]) -> Vec<u8> {
    deflate_bytes_zlib_conf(input, Compression::Default)
}
/// src/lib.rs:242-267
#[cfg(feature = "gzip")]
pub fn deflate_bytes_gzip_conf<O: Into<CompressionOptions>>(input: &[u8], options: O, gzip_header: gzip_header::GzBuilder) -> Vec<u8> {
    let h = gzip_header.into_header();
    let o = c_opts(options.into(), 2);
    unsafe {
        let cap = mi355_deflate_bound_ex(input.len(), 2, h.len(), 0);
        let mut out = Vec::<u8>::with_capacity(cap);
        let mut n = 0usize;
        if input.len() >= MULTI_GPU_FROM {
            if let Some(m) = multi() {
                let g = m.lock().unwrap();
                let rc = mi355_deflate_encode_multi(g.0, input.as_ptr(), input.len(), &o, h.as_ptr(), h.len(), out.as_mut_ptr(), cap, &mut n);
                if rc == 0 {
                    out.set_len(n);
                    return out;
                }
            }
        }
        let rc = mi355_deflate_encode_gzip(std::ptr::null_mut(), input.as_ptr(), input.len(), &o, h.as_ptr(), h.len(),
                                           out.as_mut_ptr(), cap, &mut n);
        assert!(rc == 0, "mi355_deflate_encode_gzip failed: {}", rc);
        out.set_len(n);
        out
    }
}
/// src/lib.rs:283-285
#[cfg(feature = "gzip")]
pub fn deflate_bytes_gzip(input: &[u8]) -> Vec<u8> {
    deflate_bytes_gzip_conf(input, Compression::Default, gzip_header::GzBuilder::new())
}

pub mod write {
    use super::*;

    fn err(what: &str, rc: c_int) -> io::Error {
        io::Error::new(io::ErrorKind::Other, format!("{}: mi355 error {}", what, rc))
    }

    /// Hand what the encoder has produced to the inner writer.  `W::write` may take fewer bytes than
    /// offered (tests/test.rs:163-200 issue_47): what it took is consumed, the rest is offered again --
    /// the loop of src/compress.rs:96-124 / src/writer.rs:40-47 with the Interrupted hand-shake folded in.
    fn drain<W: Write>(s: *mut Stream, w: &mut W) -> io::Result<()> {
        loop {
            let (mut p, mut n) = (std::ptr::null(), 0usize);
            unsafe { mi355_deflate_stream_output(s, &mut p, &mut n) };
            if n == 0 {
                return Ok(());
            }
            let took = match w.write(unsafe { std::slice::from_raw_parts(p, n) }) {
                Ok(0) => return Err(io::Error::new(io::ErrorKind::WriteZero, "inner writer took no bytes")),
                Ok(k) => k,
                Err(ref e) if e.kind() == io::ErrorKind::Interrupted => continue,
                Err(e) => return Err(e),
            };
            let mut k = 0usize;
            let mut scratch = vec![0u8; took];
            unsafe { mi355_deflate_stream_take_output(s, scratch.as_mut_ptr(), took, &mut k) };
        }
    }

    macro_rules! encoder {
        ($name:ident, $wrapper:expr, $doc:expr) => {
            #[doc = $doc]
            pub struct $name<W: Write> {
                s: *mut Stream,
                inner: Option<W>,
            }
            impl<W: Write> $name<W> {
                pub fn new<O: Into<CompressionOptions>>(writer: W, options: O) -> Self {
                    let o = c_opts(options.into(), $wrapper);
                    let mut s = std::ptr::null_mut();
                    let rc = unsafe { mi355_deflate_stream_new(std::ptr::null_mut(), &o, &mut s) };
                    assert!(rc == 0, "mi355_deflate_stream_new failed: {}", rc);
                    Self { s, inner: Some(writer) }
                }
                fn output_all(&mut self) -> io::Result<()> {
                    let rc = unsafe { mi355_deflate_stream_finish(self.s) };
                    if rc != 0 && rc != -6 {
                        return Err(err("finish", rc)); // (-6 = already finished)
                    }
                    drain(self.s, self.inner.as_mut().expect("writer"))
                }
                /// finish(self) -> io::Result<W>
                pub fn finish(mut self) -> io::Result<W> {
                    self.output_all()?;
                    Ok(self.inner.take().expect("writer"))
                }
                /// reset(&mut self, W) -> io::Result<W>: the finished stream goes to the old writer
                pub fn reset(&mut self, w: W) -> io::Result<W> {
                    unsafe {
                        let (mut p, mut n) = (std::ptr::null(), 0usize);
                        let rc = mi355_deflate_stream_reset(self.s, &mut p, &mut n);
                        if rc != 0 {
                            return Err(err("reset", rc));
                        }
                        let mut old = self.inner.replace(w).expect("writer");
                        old.write_all(std::slice::from_raw_parts(p, n))?;
                        Ok(old)
                    }
                }
                /// checksum() of {Zlib,Gz}Encoder
                pub fn checksum(&self) -> u32 {
                    let mut a = 0u32;
                    unsafe { mi355_deflate_stream_checksum(self.s, &mut a) };
                    a
                }
            }
            impl<W: Write> Write for $name<W> {
                fn write(&mut self, buf: &[u8]) -> io::Result<usize> {
                    let rc = unsafe { mi355_deflate_stream_write(self.s, buf.as_ptr(), buf.len()) };
                    if rc != 0 {
                        return Err(err("write", rc));
                    }
                    drain(self.s, self.inner.as_mut().expect("writer"))?; // (the header at the first write; the ranges of a stream that has not been flushed, as they are encoded)
                    Ok(buf.len())
                }
                /// Flush::Sync (src/writer.rs:134-137): the inner writer holds everything up to and including
                /// 00 00 FF FF when this returns (src/writer.rs:570-595)
                fn flush(&mut self) -> io::Result<()> {
                    let rc = unsafe { mi355_deflate_stream_flush(self.s) };
                    if rc != 0 {
                        return Err(err("flush", rc));
                    }
                    drain(self.s, self.inner.as_mut().expect("writer"))
                }
            }
            impl<W: Write> Drop for $name<W> {
                /// src/writer.rs:139-152: an encoder that is dropped unfinished finishes silently
                fn drop(&mut self) {
                    if self.inner.is_some() && !std::thread::panicking() {
                        let _ = self.output_all();
                    }
                    unsafe { mi355_deflate_stream_free(self.s) }
                }
            }
        };
    }
    encoder!(DeflateEncoder, 0, "src/writer.rs:89-152");
    encoder!(ZlibEncoder, 1, "src/writer.rs:183-290");

    #[cfg(feature = "gzip")]
    pub mod gzip {
        use super::*;
        encoder!(GzEncoder, 2, "src/writer.rs:331-467");
        impl<W: Write> GzEncoder<W> {
            /// src/writer.rs:346-358: the header is built by the real gzip-header crate, handed over as bytes
            pub fn from_builder<O: Into<CompressionOptions>>(builder: gzip_header::GzBuilder, writer: W, options: O) -> Self {
                let e = Self::new(writer, options);
                let h = builder.into_header();
                assert!(unsafe { mi355_deflate_stream_gzip_header(e.s, h.as_ptr(), h.len()) } == 0);
                e
            }
            /// src/writer.rs:393-402
            pub fn reset_with_builder(&mut self, writer: W, builder: gzip_header::GzBuilder) -> io::Result<W> {
                let old = self.reset(writer)?;
                let h = builder.into_header();
                assert!(unsafe { mi355_deflate_stream_gzip_header(self.s, h.as_ptr(), h.len()) } == 0);
                Ok(old)
            }
        }
    }
}

#[cfg(test)]
mod test {
    use super::*;
    use std::io::Write;

    /// src/lib.rs:382-391
    #[test]
    fn deflate_short() {
        assert_eq!(deflate_bytes(&[10, 10, 10, 10, 10, 55]).len(), 5);
    }
    /// src/writer.rs:570-595
    #[test]
    fn writer_sync() {
        let data = vec![7u8; 100_000];
        let mut e = write::DeflateEncoder::new(Vec::new(), CompressionOptions::default());
        e.write_all(&data[..50_000]).unwrap();
        e.flush().unwrap();
        let z = e.finish().unwrap();
        assert!(z.windows(4).any(|w| w == [0, 0, 255, 255]));
    }
}
