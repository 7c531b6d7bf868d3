"""Oracle (CPU) checks for the gzip wrapper and reset(): SURVEY section 8 f2/f4.  The gzip framing is
pinned by RFC 1952 through Python's gzip/zlib (the crate gzip-header is not in the reference tree);
reset() by the reference's own tests writer_reset / writer_reset_zlib (src/writer.rs:537-571):
the stream after a reset equals the stream of a fresh encoder."""
import gzip
import io
import os
import zlib

import datagen
import oracle_binding as ob

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_inputs")
BLANK = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff])
COMMENT = bytes([0x1f, 0x8b, 8, 16, 0, 0, 0, 0, 0, 0xff]) + b"Comment\0"


def test_crc32_matches_zlib():
    for n in (0, 1, 3, 4, 5, 511, 512, 513, 70000):
        d = datagen.rng_bytes(n, n + 1)
        assert ob.crc32(d) == zlib.crc32(d)
    d = datagen.rng_bytes(1000, 3)
    assert ob.crc32(d[500:], ob.crc32(d[:500])) == zlib.crc32(d)


def test_gzip_one_shot_is_a_gzip_member():
    for data in (b"", b"a", open(os.path.join(FIX, "pg11.txt"), "rb").read(), datagen.rng_bytes(70000, 3)):
        for hdr in (BLANK, COMMENT):
            z = ob.encode_gzip(data, hdr, level=ob.DEFAULT)
            assert z[:len(hdr)] == hdr
            assert gzip.decompress(z) == data                       # header, CRC-32 and ISIZE all checked by gzip
            assert z[len(hdr):-8] == ob.encode(data, level=ob.DEFAULT)  # lib.rs:254-256: the raw stream in between
            assert z[-8:-4] == zlib.crc32(data).to_bytes(4, "little")
            assert z[-4:] == (len(data) & 0xffffffff).to_bytes(4, "little")


# src/writer.rs:473-491 gzip_writer: two writes, a comment in the header
def test_gzip_writer():
    data = open(os.path.join(FIX, "pg11.txt"), "rb").read()
    s = ob.Stream(ob.preset(ob.DEFAULT, 2))
    s.gzip_header(COMMENT)
    s.write_all(data[:len(data) // 2])
    s.write_all(data[len(data) // 2:])
    assert s.checksum() == zlib.crc32(data)
    z = s.finish()
    assert z == ob.encode_gzip(data, COMMENT, level=ob.DEFAULT)
    f = gzip.GzipFile(fileobj=io.BytesIO(z))
    assert f.read() == data


# src/writer.rs:537-571 writer_reset, writer_reset_zlib (+ the gzip form)
def test_reset_gives_the_stream_of_a_fresh_encoder():
    data = open(os.path.join(FIX, "pg11.txt"), "rb").read()
    for wrapper in (0, 1, 2):
        s = ob.Stream(ob.preset(ob.DEFAULT, wrapper))
        if wrapper == 2:
            s.gzip_header(BLANK)
        s.write_all(data)
        res1 = s.reset()
        if wrapper == 2:
            s.gzip_header(BLANK)
        s.write_all(data)
        res2 = s.finish()
        assert res1 == res2
        fresh = ob.encode_gzip(data, BLANK) if wrapper == 2 else ob.encode(data, level=ob.DEFAULT, wrapper=wrapper)
        assert res1 == fresh
    # a reset right after a flush, and different data afterwards
    s = ob.Stream(ob.preset(ob.DEFAULT, 0))
    s.write_all(data[:50000])
    s.flush()
    a = s.reset()
    s.write_all(data[50000:])
    b = s.finish()
    assert zlib.decompressobj(-15).decompress(a) == data[:50000]
    assert b == ob.encode(data[50000:], level=ob.DEFAULT)


# the header builder of the Python mirror (no GPU needed): what Python's gzip reads back
def test_gzip_header_builder_of_the_mirror():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deflate-rs_amd"))
    import deflate_amd as da
    assert da.gzip_header() == da.BLANK_GZIP_HEADER == BLANK
    assert da.gzip_header(comment=b"Comment") == COMMENT
    data = b"This is some test data" * 10
    hdr = da.gzip_header(filename=b"name.txt", comment=b"a comment", extra=b"XY\x02\x00ab", mtime=1234567)
    z = ob.encode_gzip(data, hdr, level=ob.DEFAULT)
    f = gzip.GzipFile(fileobj=io.BytesIO(z))
    assert f.read() == data
    assert f.mtime == 1234567
