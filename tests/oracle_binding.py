"""ctypes binding of the CPU oracle (oracle/libdeflref.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product path (deflate-rs_amd/) never does.
"""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libdeflref.so")

E_REF_PANIC = -100


class Opts(C.Structure):
    _fields_ = [
        ("max_hash_checks", C.c_uint16),
        ("lazy_if_less_than", C.c_uint16),
        ("matching_type", C.c_uint8),
        ("wrapper", C.c_uint8),
    ]


class BlockInfo(C.Structure):
    _fields_ = [
        ("btype", C.c_uint8),
        ("bfinal", C.c_uint8),
        ("n_lz", C.c_uint32),
        ("in_bytes", C.c_uint64),
        ("bit_start", C.c_uint64),
    ]


FAST, DEFAULT, BEST, RLE, HUFFMAN_ONLY = 0, 1, 2, 3, 4


def build():
    src = os.path.join(ORACLE_DIR, "deflref.cpp")
    hdr = os.path.join(ORACLE_DIR, "deflref.h")
    if (not os.path.exists(LIB_PATH)) or any(
        os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(LIB_PATH) for f in (src, hdr)
    ):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        u8p = C.POINTER(C.c_uint8)
        L.deflref_preset.argtypes = [C.c_int, C.POINTER(Opts)]
        L.deflref_encode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Opts), u8p, C.c_size_t,
                                     C.POINTER(C.c_size_t)]
        L.deflref_encode.restype = C.c_int
        L.deflref_bound.argtypes = [C.c_size_t]
        L.deflref_bound.restype = C.c_size_t
        L.deflref_last_panic.restype = C.c_char_p
        L.deflref_last_hazards.restype = C.c_int
        L.deflref_stream_new.argtypes = [C.POINTER(Opts)]
        L.deflref_stream_new.restype = C.c_void_p
        L.deflref_stream_write.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.deflref_stream_flush.argtypes = [C.c_void_p]
        L.deflref_stream_finish.argtypes = [C.c_void_p]
        L.deflref_stream_output.argtypes = [C.c_void_p, C.POINTER(u8p)]
        L.deflref_stream_output.restype = C.c_size_t
        L.deflref_stream_checksum.argtypes = [C.c_void_p]
        L.deflref_stream_checksum.restype = C.c_uint32
        L.deflref_stream_free.argtypes = [C.c_void_p]
        L.deflref_trace_blocks.argtypes = [C.POINTER(BlockInfo), C.c_size_t]
        L.deflref_trace_blocks.restype = C.c_size_t
        L.deflref_lz77.argtypes = [C.c_char_p, C.c_size_t, C.c_uint16, C.c_uint16, C.c_int,
                                   C.POINTER(C.c_uint32), C.c_size_t]
        L.deflref_lz77.restype = C.c_long
        L.deflref_compress_fixed.argtypes = [C.c_char_p, C.c_size_t, u8p, C.c_size_t,
                                             C.POINTER(C.c_size_t)]
        L.deflref_longest_match.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                            C.c_size_t, C.c_uint16, C.POINTER(C.c_uint32),
                                            C.POINTER(C.c_uint32)]
        L.deflref_longest_match.restype = None
        L.deflref_get_match_length.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t]
        L.deflref_get_match_length.restype = C.c_size_t
        L.deflref_huffman_lengths.argtypes = [C.POINTER(C.c_uint16), C.c_size_t, C.c_size_t, u8p]
        L.deflref_huffman_lengths.restype = None
        L.deflref_encode_lengths.argtypes = [u8p, C.c_size_t, C.POINTER(C.c_uint16), C.c_size_t,
                                             C.POINTER(C.c_uint16)]
        L.deflref_encode_lengths.restype = C.c_long
        L.deflref_reverse_bits.argtypes = [C.c_uint16, C.c_uint8]
        L.deflref_reverse_bits.restype = C.c_uint16
        L.deflref_lsb_write.argtypes = [C.POINTER(C.c_uint16), u8p, C.c_size_t, u8p, C.c_size_t]
        L.deflref_lsb_write.restype = C.c_long
        L.deflref_stored_padding.argtypes = [C.c_uint8]
        L.deflref_stored_padding.restype = C.c_uint64
        L.deflref_get_length_code.argtypes = [C.c_uint16]
        L.deflref_get_length_code.restype = C.c_size_t
        L.deflref_get_distance_code.argtypes = [C.c_uint16]
        L.deflref_get_distance_code.restype = C.c_uint8
        L.deflref_length_extra.argtypes = [C.c_uint8, C.POINTER(C.c_uint16), u8p,
                                           C.POINTER(C.c_uint16)]
        L.deflref_length_extra.restype = None
        L.deflref_distance_extra.argtypes = [C.c_uint16, C.POINTER(C.c_uint16), u8p,
                                             C.POINTER(C.c_uint16)]
        L.deflref_distance_extra.restype = None
        L.deflref_fixed_code.argtypes = [C.c_int, C.c_uint, C.POINTER(C.c_uint16), u8p]
        L.deflref_fixed_code.restype = None
        L.deflref_zlib_header.argtypes = [C.c_uint8, u8p]
        L.deflref_zlib_header.restype = None
        L.deflref_adler32.argtypes = [C.c_char_p, C.c_size_t]
        L.deflref_adler32.restype = C.c_uint32
        L.deflref_rle_chunk.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                        C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_size_t)]
        L.deflref_rle_chunk.restype = C.c_long
        L.deflref_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
        L.deflref_crc32.restype = C.c_uint32
        L.deflref_encode_gzip.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Opts), C.c_char_p, C.c_size_t, u8p,
                                          C.c_size_t, C.POINTER(C.c_size_t)]
        L.deflref_stream_gzip_header.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.deflref_stream_reset.argtypes = [C.c_void_p, C.POINTER(u8p), C.POINTER(C.c_size_t)]
        _lib = L
    return _lib


class RefPanic(Exception):
    pass


def preset(level, wrapper=0):
    o = Opts()
    lib().deflref_preset(level, C.byref(o))
    o.wrapper = wrapper
    return o


def make_opts(max_hash_checks, lazy_if_less_than, matching_type, wrapper=0):
    return Opts(max_hash_checks, lazy_if_less_than, matching_type, wrapper)


def encode(data: bytes, opts=None, level=DEFAULT, wrapper=0) -> bytes:
    """deflate_bytes_conf / deflate_bytes_zlib_conf (src/lib.rs:137-198)."""
    if opts is None:
        opts = preset(level, wrapper)
    L = lib()
    cap = L.deflref_bound(len(data))
    out = (C.c_uint8 * cap)()
    n = C.c_size_t(0)
    rc = L.deflref_encode(bytes(data), len(data), C.byref(opts), out, cap, C.byref(n))
    if rc == E_REF_PANIC:
        raise RefPanic(L.deflref_last_panic().decode())
    if rc != 0:
        raise RuntimeError("deflref_encode rc=%d" % rc)
    return bytes(memoryview(out)[: n.value])


def encode_gzip(data: bytes, header: bytes, opts=None, level=DEFAULT) -> bytes:
    """deflate_bytes_gzip_conf (src/lib.rs:242-267); header = GzBuilder::into_header() bytes."""
    if opts is None:
        opts = preset(level, 0)
    L = lib()
    cap = L.deflref_bound(len(data)) + len(header) + 16
    out = (C.c_uint8 * cap)()
    n = C.c_size_t(0)
    rc = L.deflref_encode_gzip(bytes(data), len(data), C.byref(opts), bytes(header), len(header), out, cap, C.byref(n))
    if rc == E_REF_PANIC:
        raise RefPanic(L.deflref_last_panic().decode())
    if rc != 0:
        raise RuntimeError("deflref_encode_gzip rc=%d" % rc)
    return bytes(memoryview(out)[: n.value])


def crc32(data: bytes, crc=0) -> int:
    return lib().deflref_crc32(crc, bytes(data), len(data))


def last_hazards():
    return lib().deflref_last_hazards()


def trace_blocks():
    L = lib()
    n = L.deflref_trace_blocks(None, 0)
    arr = (BlockInfo * max(n, 1))()
    L.deflref_trace_blocks(arr, n)
    return [dict(btype=a.btype, bfinal=a.bfinal, n_lz=a.n_lz, in_bytes=a.in_bytes,
                 bit_start=a.bit_start) for a in arr[:n]]


class Stream:
    """write::{DeflateEncoder,ZlibEncoder}<Vec<u8>> (src/writer.rs:89-290)."""

    def __init__(self, opts):
        self._o = opts
        self._s = lib().deflref_stream_new(C.byref(opts))

    def _chk(self, rc):
        if rc == E_REF_PANIC:
            raise RefPanic(lib().deflref_last_panic().decode())
        if rc != 0:
            raise RuntimeError("stream rc=%d" % rc)

    def write_all(self, data: bytes):
        self._chk(lib().deflref_stream_write(self._s, bytes(data), len(data)))

    def flush(self):
        self._chk(lib().deflref_stream_flush(self._s))

    def gzip_header(self, header: bytes):
        self._chk(lib().deflref_stream_gzip_header(self._s, bytes(header), len(header)))

    def reset(self) -> bytes:
        """reset(): the bytes of the stream so far (finished); the encoder starts over"""
        p = C.POINTER(C.c_uint8)()
        n = C.c_size_t(0)
        self._chk(lib().deflref_stream_reset(self._s, C.byref(p), C.byref(n)))
        return bytes(C.string_at(p, n.value)) if n.value else b""

    def finish(self) -> bytes:
        self._chk(lib().deflref_stream_finish(self._s))
        return self.output()

    def output(self) -> bytes:
        p = C.POINTER(C.c_uint8)()
        n = lib().deflref_stream_output(self._s, C.byref(p))
        if not n:
            return b""
        if n < (1 << 31):
            return bytes(C.string_at(p, n))
        # (ctypes.string_at takes a C int: the 8 GiB stream of config 5 is 2.9 GB)
        return bytes((C.c_uint8 * n).from_address(C.addressof(p.contents)))

    def output_sha256(self):
        """(length, SHA-256) of the stream so far without a copy of it"""
        import hashlib
        p = C.POINTER(C.c_uint8)()
        n = lib().deflref_stream_output(self._s, C.byref(p))
        h = hashlib.sha256()
        step = 1 << 28
        for i in range(0, n, step):
            k = min(step, n - i)
            h.update((C.c_uint8 * k).from_address(C.addressof(p.contents) + i))
        return n, h.hexdigest()

    def checksum(self):
        return lib().deflref_stream_checksum(self._s)

    def __del__(self):
        if getattr(self, "_s", None):
            lib().deflref_stream_free(self._s)
            self._s = None


def lz77(data: bytes, max_hash_checks=1768, lazy_if_less_than=128, matching_type=1):
    """lz77_compress_conf test helper: list of ('lit', byte) / ('ld', length, distance)."""
    cap = len(data) + 16
    out = (C.c_uint32 * cap)()
    n = lib().deflref_lz77(bytes(data), len(data), max_hash_checks, lazy_if_less_than,
                           matching_type, out, cap)
    if n < 0:
        raise RefPanic(lib().deflref_last_panic().decode())
    res = []
    for v in out[:n]:
        d = v >> 16
        res.append(("lit", v & 0xFF) if d == 0 else ("ld", (v & 0xFF) + 3, d))
    return res


def compress_fixed(data: bytes) -> bytes:
    cap = len(data) * 2 + 64
    out = (C.c_uint8 * cap)()
    n = C.c_size_t(0)
    rc = lib().deflref_compress_fixed(bytes(data), len(data), out, cap, C.byref(n))
    if rc != 0:
        raise RuntimeError("rc=%d" % rc)
    return bytes(memoryview(out)[: n.value])


def longest_match(data: bytes, fill_n, position, prev_length, max_hash_checks):
    ln, d = C.c_uint32(0), C.c_uint32(0)
    lib().deflref_longest_match(bytes(data), len(data), fill_n, position, prev_length,
                                max_hash_checks, C.byref(ln), C.byref(d))
    return ln.value, d.value


def get_match_length(data: bytes, cur, check):
    return lib().deflref_get_match_length(bytes(data), len(data), cur, check)


def huffman_lengths(freqs, max_len):
    n = len(freqs)
    f = (C.c_uint16 * n)(*freqs)
    out = (C.c_uint8 * n)()
    lib().deflref_huffman_lengths(f, n, max_len, out)
    return list(out)


def encode_lengths(lens):
    n = len(lens)
    a = (C.c_uint8 * n)(*lens)
    out = (C.c_uint16 * (n + 4))()
    fr = (C.c_uint16 * 19)()
    k = lib().deflref_encode_lengths(a, n, out, n + 4, fr)
    if k < 0:
        raise RefPanic(lib().deflref_last_panic().decode())
    kinds = {0: "lit", 1: "copy", 2: "zero3", 3: "zero7"}
    return [(kinds[v >> 8], v & 0xFF) for v in out[:k]], list(fr)


def reverse_bits(n, length):
    return lib().deflref_reverse_bits(n, length)


def lsb_write(pairs):
    n = len(pairs)
    v = (C.c_uint16 * n)(*[p[0] for p in pairs])
    b = (C.c_uint8 * n)(*[p[1] for p in pairs])
    out = (C.c_uint8 * (2 * n + 16))()
    k = lib().deflref_lsb_write(v, b, n, out, 2 * n + 16)
    return list(out[:k])


def stored_padding(p):
    return lib().deflref_stored_padding(p)


def length_extra(stored_length):
    c, nb, v = C.c_uint16(), C.c_uint8(), C.c_uint16()
    lib().deflref_length_extra(stored_length, C.byref(c), C.byref(nb), C.byref(v))
    return c.value, nb.value, v.value


def distance_extra(distance):
    c, nb, v = C.c_uint16(), C.c_uint8(), C.c_uint16()
    lib().deflref_distance_extra(distance, C.byref(c), C.byref(nb), C.byref(v))
    return c.value, nb.value, v.value


def fixed_code(is_distance, symbol):
    c, ln = C.c_uint16(), C.c_uint8()
    lib().deflref_fixed_code(is_distance, symbol, C.byref(c), C.byref(ln))
    return c.value, ln.value


def zlib_header(level_bits):
    out = (C.c_uint8 * 2)()
    lib().deflref_zlib_header(level_bits, out)
    return bytes(out)


def adler32(data: bytes):
    return lib().deflref_adler32(bytes(data), len(data))


def rle_chunk(data: bytes, start, end):
    cap = len(data) + 4
    out = (C.c_uint32 * cap)()
    ov = C.c_size_t(0)
    n = lib().deflref_rle_chunk(bytes(data), len(data), start, end, out, cap, C.byref(ov))
    res = []
    for v in out[:n]:
        d = v >> 16
        res.append(("lit", v & 0xFF) if d == 0 else ("ld", (v & 0xFF) + 3, d))
    return res, ov.value
