"""Python host side of the MI355X DEFLATE encode path: a ctypes binding of libmi355deflate.so
(include/mi355_deflate.h) shaped like the reference's public API so that tests read like the
reference's own (src/lib.rs:137-216, src/writer.rs:89-290, src/compression_options.rs).

There is no CPU fallback here.  Importing works anywhere (so the symbol check can run on a
machine without a GPU); every encode call goes through the HIP kernels and raises if the library
or a gfx950 device is missing.
"""
import ctypes as C
import enum
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355_DEFLATE_LIB", os.path.join(_HERE, "libmi355deflate.so"))

FLUSH_FINISH, FLUSH_SYNC = 0, 1
OK, E_ARG, E_OUT_TOO_SMALL, E_HIP, E_UNSUPPORTED, E_REF_PANIC, E_STATE = 0, -1, -2, -3, -4, -5, -6
COMPAT_Q13 = 1

STAGES = ["links", "match", "parse", "blocks", "pack", "other"]


class Opts(C.Structure):
    _fields_ = [("max_hash_checks", C.c_uint16), ("lazy_if_less_than", C.c_uint16),
                ("matching_type", C.c_uint8), ("wrapper", C.c_uint8), ("compat", C.c_uint8),
                ("flush", C.c_uint8)]


class Info(C.Structure):
    _fields_ = [("in_len", C.c_uint64), ("out_len", C.c_uint64), ("n_tokens", C.c_uint64),
                ("n_blocks", C.c_uint32), ("n_stored", C.c_uint32), ("n_fixed", C.c_uint32),
                ("n_dynamic", C.c_uint32), ("q1_rewarm", C.c_uint32), ("q13_hits", C.c_uint32),
                ("passes", C.c_uint32), ("spec_fallback", C.c_uint32), ("stage_ms", C.c_float * 6),
                ("total_ms", C.c_float), ("match_launches", C.c_uint32), ("match_ms", C.c_float),
                ("spec_repaired", C.c_uint32), ("host_path", C.c_uint32)]


class BlockInfo(C.Structure):
    _fields_ = [("btype", C.c_uint32), ("bfinal", C.c_uint32), ("n_tokens", C.c_uint32), ("reserved", C.c_uint32),
                ("in_bytes", C.c_uint64), ("bit_start", C.c_uint64)]


class BlockCost(C.Structure):
    _fields_ = [("dyn_bits", C.c_uint64), ("dyn_est", C.c_uint64), ("static_est", C.c_uint64),
                ("fixed_bits", C.c_uint64), ("in_bytes", C.c_uint64), ("q13", C.c_uint32), ("reserved", C.c_uint32)]


class MatchingType(enum.IntEnum):
    """src/lz77.rs:27-37"""
    Greedy = 0
    Lazy = 1


class Compression(enum.IntEnum):
    """src/compression_options.rs:31-42"""
    Fast = 0
    Default = 1
    Best = 2


class CompressionOptions:
    """src/compression_options.rs:78-120; profiles :126-178."""

    def __init__(self, max_hash_checks=128, lazy_if_less_than=32, matching_type=MatchingType.Lazy):
        self.max_hash_checks = max_hash_checks
        self.lazy_if_less_than = lazy_if_less_than
        self.matching_type = MatchingType(matching_type)

    @staticmethod
    def default():
        return CompressionOptions(128, 32, MatchingType.Lazy)

    @staticmethod
    def high():
        return CompressionOptions(1768, 128, MatchingType.Lazy)

    @staticmethod
    def fast():
        return CompressionOptions(1, 0, MatchingType.Greedy)

    @staticmethod
    def huffman_only():
        return CompressionOptions(0, 0, MatchingType.Greedy)

    @staticmethod
    def rle():
        return CompressionOptions(0, 0, MatchingType.Lazy)

    @staticmethod
    def from_(o):
        """impl From<Compression> for CompressionOptions (:188-196)"""
        if isinstance(o, CompressionOptions):
            return o
        o = Compression(o)
        return {Compression.Fast: CompressionOptions.fast, Compression.Default: CompressionOptions.default,
                Compression.Best: CompressionOptions.high}[o]()

    def to_c(self, wrapper=0, compat=0, flush=0):
        return Opts(self.max_hash_checks, self.lazy_if_less_than, int(self.matching_type), wrapper, compat, flush)


class DeflateError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mi355_deflate error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Load libmi355deflate.so.  Raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libmi355deflate.so is missing (%s); run __graft_entry__.build() or "
                          "`make -C deflate-rs_amd` -- there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    u8p = C.POINTER(C.c_uint8)
    L.mi355_deflate_version.restype = C.c_int
    L.mi355_deflate_bound.argtypes = [C.c_size_t]
    L.mi355_deflate_bound.restype = C.c_size_t
    L.mi355_deflate_preset.argtypes = [C.c_int, C.POINTER(Opts)]
    L.mi355_deflate_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.mi355_deflate_ctx_destroy.argtypes = [C.c_void_p]
    L.mi355_deflate_ctx_destroy.restype = None
    L.mi355_deflate_last_error.argtypes = [C.c_void_p]
    L.mi355_deflate_last_error.restype = C.c_char_p
    L.mi355_deflate_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(Opts), u8p, C.c_size_t,
                                       C.POINTER(C.c_size_t)]
    L.mi355_deflate_encode_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Opts), C.c_void_p,
                                              C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
    L.mi355_deflate_last_info.argtypes = [C.c_void_p, C.POINTER(Info)]
    L.mi355_deflate_last_blocks.argtypes = [C.c_void_p, C.POINTER(BlockInfo), C.c_size_t, C.POINTER(C.c_size_t)]
    L.mi355_shard_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint64,
                                    C.c_uint64, C.POINTER(Opts), C.c_void_p, C.POINTER(C.c_void_p)]
    L.mi355_shard_exit_table.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.mi355_shard_spec.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.mi355_shard_emit.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_void_p)]
    L.mi355_shard_blocks.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                     C.POINTER(BlockCost), C.c_size_t]
    L.mi355_shard_blocks_ex.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64),
                                        C.POINTER(BlockCost), C.c_size_t]
    L.mi355_plan_blocks.argtypes = [C.POINTER(BlockCost), C.c_size_t, C.c_uint32, C.POINTER(BlockInfo),
                                    C.POINTER(C.c_uint64)]
    L.mi355_shard_pack.argtypes = [C.c_void_p, C.POINTER(BlockInfo), C.c_uint64, C.c_void_p, C.c_size_t,
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)]
    L.mi355_shard_end.argtypes = [C.c_void_p]
    L.mi355_shard_end.restype = None
    L.mi355_adler32_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.c_void_p]
    L.mi355_deflate_ctx_reserve.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    L.mi355_crc32_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.c_void_p]
    L.mi355_deflate_encode_gzip.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(Opts), C.c_char_p, C.c_size_t,
                                            u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mi355_deflate_encode_device_gzip.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Opts), C.c_char_p,
                                                   C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                                   C.c_void_p]
    L.mi355_deflate_stream_gzip_header.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.mi355_deflate_stream_reset.argtypes = [C.c_void_p, C.POINTER(u8p), C.POINTER(C.c_size_t)]
    L.mi355_deflate_stream_new.argtypes = [C.c_void_p, C.POINTER(Opts), C.POINTER(C.c_void_p)]
    L.mi355_deflate_stream_write.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.mi355_deflate_stream_flush.argtypes = [C.c_void_p]
    L.mi355_deflate_stream_finish.argtypes = [C.c_void_p]
    L.mi355_deflate_stream_output.argtypes = [C.c_void_p, C.POINTER(u8p), C.POINTER(C.c_size_t)]
    L.mi355_deflate_stream_take_output.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mi355_deflate_bound_ex.argtypes = [C.c_size_t, C.c_int, C.c_size_t, C.c_size_t]
    L.mi355_deflate_bound_ex.restype = C.c_size_t
    L.mi355_checksum_combine.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint64]
    L.mi355_checksum_combine.restype = C.c_uint32
    L.mi355_deflate_stream_checksum.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.mi355_deflate_stream_free.argtypes = [C.c_void_p]
    L.mi355_device_count.restype = C.c_int
    L.mi355_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    L.mi355_multi_destroy.argtypes = [C.c_void_p]
    L.mi355_multi_destroy.restype = None
    L.mi355_multi_devices.argtypes = [C.c_void_p]
    L.mi355_multi_ctx.argtypes = [C.c_void_p, C.c_int]
    L.mi355_multi_ctx.restype = C.c_void_p
    L.mi355_multi_last_error.argtypes = [C.c_void_p]
    L.mi355_multi_last_error.restype = C.c_char_p
    L.mi355_multi_layout.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_int)] + [C.POINTER(C.c_uint64)] * 4
    L.mi355_deflate_encode_multi.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(Opts), C.c_char_p, C.c_size_t, u8p,
                                             C.c_size_t, C.POINTER(C.c_size_t)]
    L.mi355_deflate_encode_multi_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(Opts), C.c_char_p,
                                                    C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mi355_multi_last_trace.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_size_t]
    L.mi355_multi_stitch_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mi355_deflate_ctx_config.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
    L.mi355_deflate_stream_held_bytes.argtypes = [C.c_void_p]
    L.mi355_deflate_stream_held_bytes.restype = C.c_uint64
    L.mi355_deflate_stream_free.restype = None
    _lib = L
    return L


EXPORTED = [
    "mi355_deflate_version", "mi355_deflate_bound", "mi355_deflate_bound_ex", "mi355_deflate_preset", "mi355_deflate_ctx_create",
    "mi355_deflate_ctx_destroy", "mi355_deflate_last_error", "mi355_deflate_encode",
    "mi355_deflate_encode_device", "mi355_deflate_last_info", "mi355_deflate_last_blocks", "mi355_adler32_device",
    "mi355_deflate_stream_new", "mi355_deflate_stream_write", "mi355_deflate_stream_flush",
    "mi355_deflate_stream_finish",
    "mi355_deflate_stream_output", "mi355_deflate_stream_take_output", "mi355_deflate_stream_checksum",
    "mi355_deflate_stream_free",
    "mi355_deflate_ctx_reserve", "mi355_deflate_stream_gzip_header", "mi355_deflate_stream_reset",
    "mi355_deflate_encode_gzip",
    "mi355_deflate_encode_device_gzip", "mi355_crc32_device",
    "mi355_shard_begin", "mi355_shard_spec", "mi355_shard_exit_table", "mi355_shard_emit", "mi355_shard_blocks", "mi355_shard_blocks_ex",
    "mi355_plan_blocks",
    "mi355_shard_pack", "mi355_shard_end", "mi355_checksum_combine",
    "mi355_deflate_ctx_config", "mi355_deflate_stream_held_bytes",
    "mi355_device_count", "mi355_multi_create", "mi355_multi_destroy", "mi355_multi_devices", "mi355_multi_ctx", "mi355_multi_last_error",
    "mi355_multi_layout", "mi355_deflate_encode_multi", "mi355_deflate_encode_multi_device", "mi355_multi_last_trace",
    "mi355_multi_stitch_info",
]


class Context:
    """One HIP device + workspace (mi355_deflate_ctx)."""

    def __init__(self, device=0):
        L = load()
        h = C.c_void_p()
        rc = L.mi355_deflate_ctx_create(device, C.byref(h))
        if rc != OK:
            raise DeflateError(rc, "cannot create a context on HIP device %d (no GPU? no CPU fallback exists)"
                               % device)
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            load().mi355_deflate_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self, rc):
        raise DeflateError(rc, load().mi355_deflate_last_error(self._h).decode())

    CFG_RANGE_BYTES, CFG_LONG_FROM, CFG_SORT_RANKS, CFG_HOST_STREAMING, CFG_MULTI_STITCH, CFG_STEPS_IN_EMIT = 1, 2, 3, 4, 5, 6
    CFG_HOST_BOUNCE, CFG_HOST_THREADS, CFG_STAGE_CLOCKS = 7, 8, 9
    HOST_PATH_PIECES, HOST_PATH_IN_THREADS, HOST_PATH_OUT_THREADS = 1, 2, 4

    def config(self, key, value):
        """mi355_deflate_ctx_config: range size / long-input threshold / where the sort takes its ranks from"""
        rc = load().mi355_deflate_ctx_config(self._h, int(key), int(value))
        if rc != OK:
            self._err(rc)

    def reserve(self, in_len, host_api=False):
        """mi355_deflate_ctx_reserve: allocate for inputs of up to in_len bytes now"""
        rc = load().mi355_deflate_ctx_reserve(self._h, in_len, 1 if host_api else 0)
        if rc != OK:
            self._err(rc)

    def encode(self, data, options=Compression.Default, wrapper=0, compat=0, flush=0):
        """Host bytes in, host bytes out (mi355_deflate_encode)."""
        L = load()
        o = CompressionOptions.from_(options).to_c(wrapper, compat, flush)
        data = bytes(data)
        cap = L.mi355_deflate_bound(len(data)) + 16
        out = (C.c_uint8 * cap)()
        n = C.c_size_t(0)
        rc = L.mi355_deflate_encode(self._h, data, len(data), C.byref(o), out, cap, C.byref(n))
        if rc != OK:
            self._err(rc)
        return bytes(memoryview(out)[: n.value])

    def encode_host_ptr(self, in_ptr, in_len, out_ptr, out_cap, options=Compression.Default, wrapper=0):
        """mi355_deflate_encode on raw host pointers (e.g. pinned buffers): H2D, encode, D2H; returns the length"""
        L = load()
        o = CompressionOptions.from_(options).to_c(wrapper, 0, 0)
        n = C.c_size_t(0)
        rc = L.mi355_deflate_encode(self._h, C.cast(C.c_void_p(in_ptr), C.c_char_p), in_len, C.byref(o),
                                    C.cast(C.c_void_p(out_ptr), C.POINTER(C.c_uint8)), out_cap, C.byref(n))
        if rc != OK:
            self._err(rc)
        return n.value

    def encode_gzip(self, data, options=Compression.Default, header=None, compat=0):
        """mi355_deflate_encode_gzip; header = GzBuilder::into_header() bytes (None: the blank one)."""
        L = load()
        o = CompressionOptions.from_(options).to_c(2, compat, 0)
        data = bytes(data)
        header = BLANK_GZIP_HEADER if header is None else bytes(header)
        cap = L.mi355_deflate_bound(len(data)) + 32 + len(header)
        out = (C.c_uint8 * cap)()
        n = C.c_size_t(0)
        rc = L.mi355_deflate_encode_gzip(self._h, data, len(data), C.byref(o), header, len(header), out, cap,
                                         C.byref(n))
        if rc != OK:
            self._err(rc)
        return bytes(memoryview(out)[: n.value])

    def crc32_device(self, d_ptr, n, stream=0):
        a = C.c_uint32(0)
        rc = load().mi355_crc32_device(self._h, C.c_void_p(d_ptr), n, C.byref(a), C.c_void_p(stream))
        if rc != OK:
            self._err(rc)
        return a.value

    def encode_device(self, d_in_ptr, in_len, d_out_ptr, out_cap, options=Compression.Default, wrapper=0,
                      compat=0, stream=0, flush=0):
        """Device pointers in/out (mi355_deflate_encode_device); returns the output length."""
        L = load()
        o = CompressionOptions.from_(options).to_c(wrapper, compat, flush)
        n = C.c_size_t(0)
        rc = L.mi355_deflate_encode_device(self._h, C.c_void_p(d_in_ptr), in_len, C.byref(o), C.c_void_p(d_out_ptr),
                                           out_cap, C.byref(n), C.c_void_p(stream))
        if rc != OK:
            self._err(rc)
        return n.value

    def info(self):
        i = Info()
        load().mi355_deflate_last_info(self._h, C.byref(i))
        d = {k: getattr(i, k) for k, _ in Info._fields_ if k not in ("stage_ms", "reserved")}
        d["stage_ms"] = {STAGES[k]: i.stage_ms[k] for k in range(6)}
        return d

    def blocks(self):
        """Block layout of the last encode, same dict shape as the oracle's trace."""
        L = load()
        n = C.c_size_t(0)
        L.mi355_deflate_last_blocks(self._h, None, 0, C.byref(n))
        arr = (BlockInfo * max(n.value, 1))()
        rc = L.mi355_deflate_last_blocks(self._h, arr, n.value, C.byref(n))
        if rc != OK:
            self._err(rc)
        return [dict(btype=a.btype, bfinal=a.bfinal, n_lz=a.n_tokens, in_bytes=a.in_bytes, bit_start=a.bit_start)
                for a in arr[: n.value]]

    def adler32_device(self, d_ptr, n, stream=0):
        a = C.c_uint32(0)
        rc = load().mi355_adler32_device(self._h, C.c_void_p(d_ptr), n, C.byref(a), C.c_void_p(stream))
        if rc != OK:
            self._err(rc)
        return a.value


_default = None


def default_context():
    global _default
    if _default is None:
        _default = Context(0)
    return _default


def bound(n):
    return load().mi355_deflate_bound(n)


# ---- the reference's one-shot functions (src/lib.rs) -------------------------------------------
def deflate_bytes_conf(data, options, ctx=None):
    """src/lib.rs:137-147"""
    return (ctx or default_context()).encode(data, options, wrapper=0)


def deflate_bytes(data, ctx=None):
    """src/lib.rs:163-165"""
    return deflate_bytes_conf(data, Compression.Default, ctx)


def deflate_bytes_zlib_conf(data, options, ctx=None):
    """src/lib.rs:182-198"""
    return (ctx or default_context()).encode(data, options, wrapper=1)


def deflate_bytes_zlib(data, ctx=None):
    """src/lib.rs:216-218"""
    return deflate_bytes_zlib_conf(data, Compression.Default, ctx)


# GzBuilder::new().into_header() of crate gzip-header 1.0 (the crate is not in the reference tree)
BLANK_GZIP_HEADER = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff])


def gzip_header(filename=None, comment=None, extra=None, mtime=0, xfl=0, os_code=255):
    """RFC 1952 member header as GzBuilder builds it (fields in the order FEXTRA, FNAME, FCOMMENT)."""
    flg = (4 if extra is not None else 0) | (8 if filename is not None else 0) | (16 if comment is not None else 0)
    h = bytearray([0x1f, 0x8b, 8, flg]) + int(mtime).to_bytes(4, "little") + bytes([xfl, os_code])
    if extra is not None:
        h += len(extra).to_bytes(2, "little") + bytes(extra)
    if filename is not None:
        h += bytes(filename) + b"\0"
    if comment is not None:
        h += bytes(comment) + b"\0"
    return bytes(h)


def deflate_bytes_gzip_conf(data, options, header=None, ctx=None):
    """src/lib.rs:242-267 (feature "gzip"); header = GzBuilder::into_header() bytes"""
    return (ctx or default_context()).encode_gzip(data, options, header)


def deflate_bytes_gzip(data, ctx=None):
    """src/lib.rs:283-285"""
    return deflate_bytes_gzip_conf(data, Compression.Default, None, ctx)


# ---- the reference's Write encoders (src/writer.rs) ---------------------------------------------
class _Encoder:
    _wrapper = 0

    def __init__(self, writer, options=Compression.Default, ctx=None):
        """::new(writer, options) (writer.rs:93-99, 189-199); `writer` needs a .write(bytes)."""
        self._ctx = ctx or default_context()
        self._w = writer
        o = CompressionOptions.from_(options).to_c(self._wrapper, 0)
        h = C.c_void_p()
        rc = load().mi355_deflate_stream_new(self._ctx._h, C.byref(o), C.byref(h))
        if rc != OK:
            raise DeflateError(rc, "stream_new")
        self._s = h

    def _drain(self):
        """Hand what the encoder has produced to the inner writer, the way the reference does
        (compress.rs:96-124, 280-299; writer.rs:40-47): `W.write` may accept fewer bytes than offered
        (tests/test.rs:163-200 SmallWriter); a writer that returns None took everything."""
        L = load()
        p = C.POINTER(C.c_uint8)()
        n = C.c_size_t(0)
        while True:
            L.mi355_deflate_stream_output(self._s, C.byref(p), C.byref(n))
            if not n.value:
                return
            chunk = C.string_at(p, n.value)
            took = self._w.write(chunk)
            took = len(chunk) if took is None else int(took)
            if took <= 0:
                raise IOError("inner writer accepted no bytes (io::ErrorKind::WriteZero)")
            k = C.c_size_t(0)
            buf = (C.c_uint8 * took)()
            L.mi355_deflate_stream_take_output(self._s, buf, took, C.byref(k))

    def write(self, buf):
        """io::Write::write (always consumes everything, like write_all)"""
        buf = bytes(buf)
        rc = load().mi355_deflate_stream_write(self._s, buf, len(buf))
        if rc != OK:
            raise DeflateError(rc, "stream_write")
        self._drain()
        return len(buf)

    write_all = write

    def flush(self):
        """io::Write::flush = Flush::Sync (writer.rs:134-137): sync marker, window kept; the inner writer
        holds the bytes -- ending in 00 00 FF FF -- when this returns (writer.rs:570-595)"""
        L = load()
        rc = L.mi355_deflate_stream_flush(self._s)
        if rc != OK:
            raise DeflateError(rc, L.mi355_deflate_last_error(self._ctx._h).decode())
        self._drain()

    def finish(self):
        """finish(self) -> W (writer.rs:103-108, 209-214)"""
        L = load()
        rc = L.mi355_deflate_stream_finish(self._s)
        if rc != OK:
            raise DeflateError(rc, L.mi355_deflate_last_error(self._ctx._h).decode())
        self._drain()
        self._done = True
        return self._w

    def reset(self, writer):
        """reset(&mut self, W) -> W (writer.rs:110-117, 216-223, 383-402): the finished stream goes to the
        old writer, which is returned; the encoder starts over on `writer` with the same options"""
        L = load()
        p = C.POINTER(C.c_uint8)()
        n = C.c_size_t(0)
        rc = L.mi355_deflate_stream_reset(self._s, C.byref(p), C.byref(n))
        if rc != OK:
            raise DeflateError(rc, L.mi355_deflate_last_error(self._ctx._h).decode())
        rest = C.string_at(p, n.value) if n.value else b""
        while rest:
            took = self._w.write(rest)
            took = len(rest) if took is None else int(took)
            if took <= 0:
                raise IOError("inner writer accepted no bytes (io::ErrorKind::WriteZero)")
            rest = rest[took:]
        old, self._w = self._w, writer
        return old

    def close(self):
        """Drop (writer.rs:139-152): an encoder that goes away unfinished finishes its stream; errors are
        swallowed, as Drop must"""
        if getattr(self, "_s", None) and not getattr(self, "_done", False):
            try:
                self.finish()
            except Exception:
                pass

    def checksum(self):
        """{Zlib,Gz}Encoder::checksum() (writer.rs:248-250, :428-430)"""
        a = C.c_uint32(0)
        rc = load().mi355_deflate_stream_checksum(self._s, C.byref(a))
        if rc != OK:
            raise DeflateError(rc, "stream_checksum")
        return a.value

    def __del__(self):
        if getattr(self, "_s", None):
            try:
                self.close()
                load().mi355_deflate_stream_free(self._s)
            except Exception:
                pass
            self._s = None


class DeflateEncoder(_Encoder):
    """write::DeflateEncoder (src/writer.rs:89-152)"""
    _wrapper = 0


class ZlibEncoder(_Encoder):
    """write::ZlibEncoder (src/writer.rs:183-290)"""
    _wrapper = 1


class GzEncoder(_Encoder):
    """write::gzip::GzEncoder (src/writer.rs:293-467, feature "gzip")"""
    _wrapper = 2

    @classmethod
    def from_builder(cls, header, writer, options=Compression.Default, ctx=None):
        """from_builder(builder, writer, options) (:346-358); header = builder.into_header() bytes"""
        e = cls(writer, options, ctx)
        e.set_header(header)
        return e

    def set_header(self, header):
        header = bytes(header)
        rc = load().mi355_deflate_stream_gzip_header(self._s, header, len(header))
        if rc != OK:
            raise DeflateError(rc, "stream_gzip_header")

    def reset_with_builder(self, writer, header):
        """reset_with_builder (:393-402)"""
        old = self.reset(writer)
        self.set_header(header)
        return old


# ---- sharded, stream-exact encode: the per-rank phases (mi355_shard_*) -------------------------------
ZONE = 576
BLOCK_TOKENS = 31744


class MultiGpu:
    """mi355_multi: one input over several GPUs of this node in one call (one process, a thread per device); the
    stream is the one a single encoder produces for the whole input.  devices: HIP device per rank (a device may be
    named more than once: the ranks then share it)."""
    TRACE = ["tables", "wait1", "tokens", "wait2", "block costs", "wait3", "plan+pack+copy", "wait4", "seams+framing",
             "host work of the exchanges", "call"]

    def __init__(self, devices=None):
        """devices=None: every device of the node"""
        h = C.c_void_p()
        if devices is None:
            rc = load().mi355_multi_create(None, 0, C.byref(h))
            devices = list(range(load().mi355_device_count()))
        else:
            arr = (C.c_int * len(devices))(*devices)
            rc = load().mi355_multi_create(arr, len(devices), C.byref(h))
        if rc != OK:
            raise DeflateError(rc, "cannot create contexts on HIP devices %r" % (list(devices),))
        self._h = h
        self.devices = list(devices)

    def close(self):
        if getattr(self, "_h", None):
            load().mi355_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self, rc):
        raise DeflateError(rc, load().mi355_multi_last_error(self._h).decode())

    def config(self, key, value):
        """mi355_deflate_ctx_config of rank 0's context (CFG_RANGE_BYTES: twice that is the most one rank takes)"""
        rc = load().mi355_deflate_ctx_config(C.c_void_p(load().mi355_multi_ctx(self._h, 0)), key, value)
        if rc != OK:
            self._err(rc)

    def layout(self, in_len, rank):
        """-> dict(n_ranks, g_lo, g_hi, lo, hi): the bytes rank `rank` holds and owns"""
        n = C.c_int(0)
        v = [C.c_uint64(0) for _ in range(4)]
        rc = load().mi355_multi_layout(self._h, in_len, rank, C.byref(n), *[C.byref(x) for x in v])
        if rc != OK:
            self._err(rc)
        return dict(n_ranks=n.value, g_lo=v[0].value, g_hi=v[1].value, lo=v[2].value, hi=v[3].value)

    def encode(self, data, options=Compression.Default, wrapper=0, compat=0, gzip_header=None):
        """deflate_bytes_conf / _zlib_conf / _gzip_conf over all devices: bytes in, bytes out"""
        data = bytes(data) if not isinstance(data, bytes) else data
        o = CompressionOptions.from_(options).to_c(wrapper, compat, FLUSH_FINISH)
        hdr = bytes(gzip_header) if gzip_header is not None else None
        cap = load().mi355_deflate_bound_ex(len(data), wrapper, len(hdr) if hdr else 10, 0) + 8
        out = (C.c_uint8 * cap)()
        n = C.c_size_t(0)
        rc = load().mi355_deflate_encode_multi(self._h, data, len(data), C.byref(o), hdr, len(hdr) if hdr else 0, out, cap, C.byref(n))
        if rc != OK:
            self._err(rc)
        return bytes(memoryview(out)[: n.value])

    def encode_host_ptr(self, in_ptr, n, out_ptr, out_cap, options=Compression.Default, wrapper=0):
        o = CompressionOptions.from_(options).to_c(wrapper, 0, FLUSH_FINISH)
        got = C.c_size_t(0)
        rc = load().mi355_deflate_encode_multi(self._h, C.cast(C.c_void_p(in_ptr), C.c_char_p), n, C.byref(o), None, 0,
                                               C.cast(C.c_void_p(out_ptr), C.POINTER(C.c_uint8)), out_cap, C.byref(got))
        if rc != OK:
            self._err(rc)
        return got.value

    def encode_device(self, d_ext_ptrs, in_len, d_out_ptr, out_cap, options=Compression.Default, wrapper=0, compat=0,
                      gzip_header=None):
        """d_ext_ptrs[r] = device pointer (on rank r's device) to the bytes layout(in_len, r) names; d_out on rank 0's"""
        o = CompressionOptions.from_(options).to_c(wrapper, compat, FLUSH_FINISH)
        hdr = bytes(gzip_header) if gzip_header is not None else None
        arr = (C.c_void_p * len(d_ext_ptrs))(*[C.c_void_p(p) for p in d_ext_ptrs])
        n = C.c_size_t(0)
        rc = load().mi355_deflate_encode_multi_device(self._h, arr, in_len, C.byref(o), hdr, len(hdr) if hdr else 0,
                                                      C.c_void_p(d_out_ptr), out_cap, C.byref(n))
        if rc != OK:
            self._err(rc)
        return n.value

    def stitch_info(self):
        """how the packed ranges of the last call reached rank 0 ("rccl" / "peer"), and the ranks RCCL reports (0: no communicator)"""
        a, b = C.c_int(0), C.c_int(0)
        load().mi355_multi_stitch_info(self._h, C.byref(a), C.byref(b))
        return {"stitch": "rccl" if a.value else "peer", "rccl_ranks": b.value}

    def trace(self):
        t = (C.c_double * 11)()
        load().mi355_multi_last_trace(self._h, t, 11)
        return dict(zip(self.TRACE, [round(x, 4) for x in t]))

    def rank_info(self, rank):
        info = Info()
        load().mi355_deflate_last_info(C.c_void_p(load().mi355_multi_ctx(self._h, rank)), C.byref(info))
        return {"match_ms": info.match_ms, "match_launches": info.match_launches}


class Shard:
    """One rank's session of the sharded (P1) encode; see include/mi355_deflate.h."""

    def __init__(self, ctx, d_ext_ptr, n_ext, parse_lo, parse_hi, global_lo, n_global, options=Compression.Default,
                 compat=0, stream=0):
        self.ctx = ctx
        o = CompressionOptions.from_(options).to_c(0, compat, 0)
        h = C.c_void_p()
        rc = load().mi355_shard_begin(ctx._h, C.c_void_p(d_ext_ptr), n_ext, parse_lo, parse_hi, global_lo, n_global,
                                      C.byref(o), C.c_void_p(stream), C.byref(h))
        if rc != OK:
            ctx._err(rc)
        self._h = h
        self.compat = compat
        self.nb = 0

    def spec(self):
        """-> (held, entry, exit) of the range's speculative parse (buffer coordinates)"""
        h, e, x = C.c_int(0), C.c_uint64(0), C.c_uint64(0)
        rc = load().mi355_shard_spec(self._h, C.byref(h), C.byref(e), C.byref(x))
        if rc != OK:
            self.ctx._err(rc)
        return bool(h.value), e.value, x.value

    def exit_table(self):
        t = (C.c_uint32 * ZONE)()
        rc = load().mi355_shard_exit_table(self._h, t)
        if rc != OK:
            self.ctx._err(rc)
        return list(t)

    def emit(self, entry):
        """-> (token count, device pointer of the dense token array)"""
        n = C.c_uint64(0)
        p = C.c_void_p()
        rc = load().mi355_shard_emit(self._h, entry, C.byref(n), C.byref(p))
        if rc != OK:
            self.ctx._err(rc)
        return n.value, (p.value or 0)

    def _blocks_call(self, skip, d_tail_ptr, n_tail, owns_final, nb, costs, cap):
        if owns_final is None:  # "the last rank owns the last block" (every range reaches its next block boundary)
            return load().mi355_shard_blocks(self._h, skip, C.c_void_p(d_tail_ptr), n_tail, C.byref(nb), costs, cap)
        return load().mi355_shard_blocks_ex(self._h, skip, C.c_void_p(d_tail_ptr), n_tail, 1 if owns_final else 0,
                                            C.byref(nb), costs, cap)

    def blocks(self, skip, d_tail_ptr, n_tail, owns_final=None):
        """-> list of cost tuples (dyn_bits, dyn_est, static_est, fixed_bits, in_bytes, q13)"""
        n, costs = self.blocks_raw(skip, d_tail_ptr, n_tail, owns_final)
        return [(c.dyn_bits, c.dyn_est, c.static_est, c.fixed_bits, c.in_bytes, c.q13) for c in costs[:n]]

    def blocks_raw(self, skip, d_tail_ptr, n_tail, owns_final=None):
        """-> (number of blocks, ctypes array of BlockCost): no per-block Python work (the distributed driver)"""
        nb = C.c_uint64(0)
        cap = 1 << 14
        costs = (BlockCost * cap)()
        rc = self._blocks_call(skip, d_tail_ptr, n_tail, owns_final, nb, costs, cap)
        if rc != OK:
            self.ctx._err(rc)
        if nb.value > cap:
            costs = (BlockCost * nb.value)()
            rc = self._blocks_call(skip, d_tail_ptr, n_tail, owns_final, nb, costs, nb.value)
            if rc != OK:
                self.ctx._err(rc)
        self.nb = nb.value
        return nb.value, costs

    def pack_raw(self, plans_arr, first, end_bit, d_out_ptr, out_cap):
        """plans_arr: ctypes array of BlockInfo for the whole stream, `first` = index of this rank's first block"""
        fb = C.c_uint64(0)
        nbts = C.c_size_t(0)
        ptr = C.cast(C.byref(plans_arr, first * C.sizeof(BlockInfo)), C.POINTER(BlockInfo))
        rc = load().mi355_shard_pack(self._h, ptr, end_bit, C.c_void_p(d_out_ptr), out_cap, C.byref(fb), C.byref(nbts))
        if rc != OK:
            self.ctx._err(rc)
        return fb.value, nbts.value

    def pack(self, plans, end_bit, d_out_ptr, out_cap):
        """plans: list of (btype, bfinal, bit_start) of this rank's blocks -> (first_byte, n_bytes)"""
        arr = (BlockInfo * max(1, len(plans)))()
        for i, (bt, bf, bs) in enumerate(plans):
            arr[i].btype, arr[i].bfinal, arr[i].bit_start = bt, bf, bs
        fb = C.c_uint64(0)
        nbts = C.c_size_t(0)
        rc = load().mi355_shard_pack(self._h, arr, end_bit, C.c_void_p(d_out_ptr), out_cap, C.byref(fb), C.byref(nbts))
        if rc != OK:
            self.ctx._err(rc)
        return fb.value, nbts.value

    def close(self):
        if getattr(self, "_h", None):
            load().mi355_shard_end(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_blocks_raw(costs_arr, n, compat=0):
    """plan_blocks over a ctypes array of BlockCost -> (ctypes array of BlockInfo, total_bits)"""
    out = (BlockInfo * max(1, n))()
    tot = C.c_uint64(0)
    rc = load().mi355_plan_blocks(costs_arr, n, compat, out, C.byref(tot))
    if rc != OK:
        raise DeflateError(rc, "mi355_plan_blocks")
    return out, tot.value


def plan_blocks(costs, compat=0):
    """The serial block plan over the costs of ALL blocks of the stream (host; identical on every rank).
    -> (list of (btype, bfinal, bit_start), total_bits)"""
    n = len(costs)
    arr = (BlockCost * max(1, n))()
    for i, c in enumerate(costs):
        (arr[i].dyn_bits, arr[i].dyn_est, arr[i].static_est, arr[i].fixed_bits, arr[i].in_bytes, arr[i].q13) = c
    out = (BlockInfo * max(1, n))()
    tot = C.c_uint64(0)
    rc = load().mi355_plan_blocks(arr, n, compat, out, C.byref(tot))
    if rc != OK:
        raise DeflateError(rc, "mi355_plan_blocks")
    return [(o.btype, o.bfinal, o.bit_start) for o in out[:n]], tot.value
