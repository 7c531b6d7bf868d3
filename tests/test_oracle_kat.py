"""Pins the CPU oracle against every known-answer test the reference's own suite holds for
the encode hot path (group K).  Each test cites the reference test it
replays.  CPU only."""
import os
import zlib

import oracle_binding as ob

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_inputs")


def inflate_raw(b):
    d = zlib.decompressobj(-15)
    out = d.decompress(b) + d.flush()
    assert d.eof, "stream not terminated"
    assert d.unused_data == b""
    return out


# src/compress.rs:333-345 fixed_example
def test_fixed_example():
    check = bytes([0x73, 0x49, 0x4D, 0xCB, 0x49, 0x2C, 0x49, 0x55, 0x00, 0x11, 0x00])
    c = ob.compress_fixed(b"Deflate late")
    assert c == check
    assert inflate_raw(c) == b"Deflate late"


# src/compress.rs:311-331 fixed_string_mem / fixed_data
def test_fixed_roundtrips():
    for data in (b"                    GNU GENERAL PUBLIC LICENSE", bytes([190]) * 400):
        assert inflate_raw(ob.compress_fixed(data)) == data


# src/bitstream.rs:131-178 write_bits
def test_lsb_writer_vector():
    inp = [(3, 3), (10, 8), (88, 7), (0, 2), (0, 5), (0, 0), (238, 8), (126, 8), (161, 8), (10, 8),
           (238, 8), (174, 8), (126, 8), (174, 8), (65, 8), (142, 8), (62, 8), (10, 8), (1, 8),
           (161, 8), (78, 8), (62, 8), (158, 8), (206, 8), (10, 8), (64, 7), (0, 0), (24, 5),
           (0, 0), (174, 8), (126, 8), (193, 8), (174, 8)]
    expected = [83, 192, 2, 220, 253, 66, 21, 220, 93, 253, 92, 131, 28, 125, 20, 2, 66, 157, 124,
                60, 157, 21, 128, 216, 213, 47, 216, 21]
    assert ob.lsb_write(inp) == expected


# src/bit_reverse.rs:16-24
def test_reverse_bits():
    assert ob.reverse_bits(0b0111_0100, 8) == 0b0010_1110
    assert ob.reverse_bits(0b1100_1100_1100_1100, 16) == 0b0011_0011_0011_0011
    # 16-bit case in the reference: reverse twice is identity
    for v in (0x1234, 0xFFFE, 1):
        assert ob.reverse_bits(ob.reverse_bits(v, 16), 16) == v


# src/huffman_table.rs:506-527 make_table_fixed
def test_fixed_table_codes():
    assert ob.fixed_code(0, 0)[0] == 0b00001100
    assert ob.fixed_code(0, 143)[0] == 0b11111101
    assert ob.fixed_code(0, 144)[0] == 0b000010011
    assert ob.fixed_code(0, 255)[0] == 0b111111111
    assert ob.fixed_code(0, 256)[0] == 0b0000000
    assert ob.fixed_code(0, 279)[0] == 0b1110100
    assert ob.fixed_code(0, 280)[0] == 0b00000011
    assert ob.fixed_code(0, 287)[0] == 0b11100011
    assert ob.fixed_code(1, 0)[0] == 0
    assert ob.fixed_code(1, 5)[0] == 20
    # get_length_distance_code(4, 5)
    code, nb, val = ob.length_extra(4 - 3)
    assert ob.fixed_code(0, code)[0] == 0b00100000
    dcode, dnb, dval = ob.distance_extra(5)
    assert ob.fixed_code(1, dcode)[0] == 0b00100
    assert dnb == 1 and dval == 0


# src/huffman_table.rs:439-459 test_get_length_code
def test_length_codes():
    assert ob.length_extra(4 - 3) == (258, 0, 0)
    assert ob.length_extra(165 - 3) == (282, 5, 2)
    assert ob.length_extra(257 - 3) == (284, 5, 30)
    c, nb, _ = ob.length_extra(258 - 3)
    assert (c, nb) == (285, 0)
    assert ob.lib().deflref_get_length_code(3) == 257
    assert ob.lib().deflref_get_length_code(258) == 285


# src/huffman_table.rs:461-485 test_distance_code / test_distance_extra_bits, :529-539
def test_distance_codes():
    g = ob.lib().deflref_get_distance_code
    assert g(1) == 0 and g(0) == 0 and g(50000) == 0
    assert g(6146) == 25 and g(256) == 15 and g(4733) == 24 and g(257) == 16
    c, nb, v = ob.distance_extra(527)
    assert (c, nb, v) == (18, 8, 0b1110)
    assert ob.distance_extra(256)[:2] == (15, 6)
    assert ob.distance_extra(4733)[:2] == (24, 11)
    extra = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12,
             12, 13, 13]
    base = [0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024,
            1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 24576]
    for code in range(30):
        d = base[code] + 1
        assert ob.distance_extra(d) == (code, extra[code], 0)
    # every distance: code/extra agree with RFC 1951 ranges
    for d in range(1, 32769):
        c = g(d)
        assert base[c] + 1 <= d and (c == 29 or d < base[c + 1] + 1)


# RFC 1951 length table (what LENGTH_CODE/BASE_LENGTH src/huffman_table.rs:50-68 encode)
def test_all_length_codes_rfc():
    base = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99,
            115, 131, 163, 195, 227, 258]
    extra = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
    for ln in range(3, 259):
        code, nb, val = ob.length_extra(ln - 3)
        k = code - 257
        assert extra[k] == nb
        assert base[k] + val == ln
        assert val < (1 << nb) or nb == 0 and val == 0


def lit(v):
    return ("lit", v)


def zero(r):
    if r <= 1:
        return ("lit", 0)
    return ("zero3", r) if r <= 10 else ("zero7", r)


def copy(c):
    return ("copy", c)


# src/length_encode.rs:440-567 test_encode_lengths
def test_encode_lengths_vectors():
    fixed = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8
    enc, fr = ob.encode_lengths(fixed)
    assert fr[0:7] == [0] * 7 and fr[10:16] == [0] * 6 and fr[17:19] == [0, 0]
    enc, _ = ob.encode_lengths([0, 0, 5, 0, 15, 1, 0, 0, 0, 2, 4, 4, 4, 4, 3, 5, 5, 5, 5])
    assert enc == [lit(0), lit(0), lit(5), lit(0), lit(15), lit(1), zero(3), lit(2), lit(4),
                   copy(3), lit(3), lit(5), copy(3)]
    enc, _ = ob.encode_lengths([0, 0, 0, 5, 2, 3, 0, 0, 0])
    assert enc == [zero(3), lit(5), lit(2), lit(3), zero(3)]
    enc, _ = ob.encode_lengths([0, 0, 0, 3, 3, 3, 5, 4, 4, 4, 4, 0, 0])
    assert enc == [zero(3), lit(3), lit(3), lit(3), lit(5), lit(4), copy(3), lit(0), lit(0)]
    lens = ([0, 0, 4, 0, 0, 4, 0, 0, 0, 0, 0, 4, 4] + [0] * 9 + [3] + [0] * 32 + [4] + [0] * 200
            + [4] + [0] * 28 + [1, 1])
    ob.encode_lengths(lens)  # must not panic
    lens = [
        0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 9, 0, 0, 9, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
        0, 0, 0, 6, 0, 0, 0, 8, 0, 0, 0, 0, 8, 0, 0, 7, 8, 7, 8, 6, 6, 8, 0, 7, 6, 7, 8, 7, 7,
        8, 0, 0, 0, 0, 0, 8, 8, 0, 8, 7, 0, 10, 8, 0, 8, 0, 10, 10, 8, 8, 10, 8, 0, 8, 7, 0,
        10, 0, 7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 7, 7, 6, 7, 8, 8, 6, 0, 0, 8, 8, 7, 8, 8, 0,
        7, 6, 6, 8, 8, 8, 10, 10] + [0] * 133 + [10, 4,
        3, 3, 4, 4, 5, 5, 5, 5, 5, 8, 8, 6, 7, 8, 10, 10, 0, 9,
        0, 0, 0, 0, 0, 0, 0, 8, 8, 8, 8, 6, 6, 5, 5, 5, 5, 6, 5, 5, 4, 4, 4, 4, 4, 4, 3, 4, 3, 4]
    enc, _ = ob.encode_lengths(lens)
    assert enc[:10] == [zero(10), lit(9), lit(0), lit(0), lit(9), zero(18), lit(6), zero(3),
                        lit(8), zero(4)]
    assert enc[10:20] == [lit(8), lit(0), lit(0), lit(7), lit(8), lit(7), lit(8), lit(6), lit(6),
                          lit(8)]
    assert ob.encode_lengths([1, 1, 1, 2])[0] == [lit(1), lit(1), lit(1), lit(2)]
    assert ob.encode_lengths([0, 0, 3])[0] == [lit(0), lit(0), lit(3)]
    assert ob.encode_lengths([0, 0, 0, 5, 2])[0] == [zero(3), lit(5), lit(2)]
    assert ob.encode_lengths([0, 0, 0, 5, 0])[0][-1] != lit(5)
    assert ob.encode_lengths([0, 4, 4, 4, 4, 0])[0][-1] == zero(0)


def expand_encoded(enc):
    out = []
    for k, v in enc:
        if k == "lit":
            out.append(v)
        elif k == "copy":
            out.extend([out[-1]] * v)
        else:
            out.extend([0] * v)
    return out


def test_encode_lengths_always_expands_back():
    import random
    rnd = random.Random(7)
    for _ in range(2000):
        n = rnd.randint(1, 320)
        mode = rnd.random()
        if mode < 0.3:
            lens = [rnd.choice([0, 0, 0, 5, 6]) for _ in range(n)]
        elif mode < 0.6:
            lens = []
            while len(lens) < n:
                lens.extend([rnd.randint(0, 15)] * rnd.randint(1, 150))
            lens = lens[:n]
        else:
            lens = [rnd.randint(0, 15) for _ in range(n)]
        enc, fr = ob.encode_lengths(lens)
        assert expand_encoded(enc) == lens
        for k, v in enc:
            if k == "copy":
                assert 3 <= v <= 6
            if k == "zero3":
                assert 3 <= v <= 10
            if k == "zero7":
                assert 11 <= v <= 138


# src/length_encode.rs:569-614 test_lengths_from_frequencies
def test_lengths_from_frequencies():
    assert ob.huffman_lengths([1, 1, 5, 7, 10, 14], 4) == [4, 4, 3, 2, 2, 2]
    assert ob.huffman_lengths([1, 5, 1, 7, 10, 14], 4) == [4, 3, 4, 2, 2, 2]
    res = ob.huffman_lengths([0, 25, 0, 10, 2, 4], 4)
    assert res[0] == 0 and res[2] == 0 and res[1] < 4
    assert ob.huffman_lengths([0, 0, 0, 0, 0, 0, 0, 0, 55, 0, 0, 0], 5) == \
        [0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0]
    assert ob.huffman_lengths([0] * 30, 5) == [0] * 30
    f = [3] * 286
    f[55] = 65535 // 3
    f[125] = 65535 // 3
    res = ob.huffman_lengths(f, 15)
    assert len(res) == 286 and res[55] < 3 and res[125] < 3


OPT_FREQS = [
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 44, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 68, 0, 14, 0, 0, 0, 0, 3, 7, 6, 1, 0, 12, 14, 9, 2, 6, 9, 4, 1, 1, 4, 1, 1, 0,
    0, 1, 3, 0, 6, 0, 0, 0, 4, 4, 1, 2, 5, 3, 2, 2, 9, 0, 0, 3, 1, 5, 5, 8, 0, 6, 10, 5, 2,
    0, 0, 1, 2, 0, 8, 11, 4, 0, 1, 3, 31, 13, 23, 22, 56, 22, 8, 11, 43, 0, 7, 33, 15, 45,
    40, 16, 1, 28, 37, 35, 26, 3, 7, 11, 9, 1, 1, 0, 1] + [0] * 131 + [
    1, 126, 114, 66, 31, 41, 25, 15, 21, 20, 16, 15, 10, 7, 5, 1, 1]


# src/length_encode.rs:616-660 optimal_lengths
def test_optimal_lengths_7701():
    assert len(OPT_FREQS) == 274 or len(OPT_FREQS) > 0
    lens = ob.huffman_lengths(OPT_FREQS, 15)
    assert sum(f * l for f, l in zip(OPT_FREQS, lens)) == 7701
    # Kraft equality for a complete code
    assert sum(2.0 ** -l for l in lens if l) == 1.0


def test_length_limited_codes_are_valid():
    import random
    rnd = random.Random(11)
    for _ in range(500):
        n = rnd.choice([19, 30, 286])
        mx = 7 if n == 19 else 15
        freqs = [rnd.choice([0, 0, 1, 2, 3, 50, 1000, rnd.randint(0, 30000)]) for _ in range(n)]
        lens = ob.huffman_lengths(freqs, mx)
        used = [l for l in lens if l]
        assert all(l <= mx for l in used)
        assert all((f > 0) == (l > 0) for f, l in zip(freqs, lens))
        if len(used) > 1:
            assert sum(2.0 ** -l for l in used) <= 1.0 + 1e-12


# src/huffman_lengths.rs:374-384 padding
def test_stored_padding():
    assert [ob.stored_padding(i) for i in range(8)] == [5, 4, 3, 2, 1, 0, 7, 6]


# src/rle.rs:82-104 rle_compress
def test_rle_vector():
    data = b"textaaaaaaaaatext"
    toks, overlap = ob.rle_chunk(data, 0, len(data))
    exp = [lit(ord(c)) for c in "texta"] + [("ld", 8, 1)] + [lit(ord(c)) for c in "text"]
    assert toks == exp and overlap == 0


# src/lz77.rs:937-947 compress_short and :970-984 lazy
def test_lz77_short_and_lazy():
    res = ob.lz77(b"Deflate late")
    assert res[-1] == ("ld", 4, 5)
    res = ob.lz77(b"nba badger nbadger")
    assert res[-1][0] == "ld" and res[-1][1] == 6


def lz_decode(toks):
    out = bytearray()
    for t in toks:
        if t[0] == "lit":
            out.append(t[1])
        else:
            for _ in range(t[1]):
                out.append(out[-t[2]])
    return bytes(out)


# src/lz77.rs:949-968 compress_long, :993-1033 exact_window_size / border*
def test_lz77_roundtrips():
    pg = open(os.path.join(FIX, "pg11.txt"), "rb").read()
    toks = ob.lz77(pg)
    assert len(toks) < len(pg)
    assert lz_decode(toks) == pg
    for data in (bytes(32768), bytes(32768) + bytes([22]) * 32768,
                 bytes(32768) + bytes([22]) * 32768 + bytes([5]) * 300,
                 bytes(range(256)) * 300):
        for mt in (0, 1):
            assert lz_decode(ob.lz77(data, 128, 32, mt)) == data


# src/matching.rs:296-343
def test_matching_kats():
    arr = bytes([5, 5, 5, 5, 5, 9, 9, 2, 3, 5, 5, 5, 5, 5])
    assert ob.get_match_length(arr, 9, 0) == 5
    assert ob.get_match_length(arr, 9, 7) == 0
    assert ob.get_match_length(arr, 10, 0) == 4
    data = b"xTest data, Test_data,zTest data"
    # filled_hash_table(&data[..23+1+3-1]); longest_match_current: position = current_head = 23,
    # prev_length = MIN_MATCH-1, MAX_HASH_CHECKS
    assert ob.longest_match(data, 26, 23, 2, 32768) == (9, 22)
    arr2 = bytes([10, 10, 10, 10, 10, 10, 10, 10, 2, 3, 5, 10, 10, 10, 10, 10])
    # filled_hash_table(&arr2[..3+1+1+2]) -> positions 0..4 inserted, all hash-equal; head = 4
    assert ob.longest_match(arr2, 7, 4, 2, 32768) == (4, 1)
    # match_index_zero
    assert ob.longest_match(b"AAAAAAA", 5, 1, 0, 4096) == (6, 1)


# src/lib.rs:382-391 deflate_short
def test_deflate_short_is_5_bytes():
    data = bytes([10, 10, 10, 10, 10, 55])
    c = ob.encode(data)
    assert len(c) == 5
    assert inflate_raw(c) == data


# tests/test.rs:58-63 block_type
def test_block_type_short_bin_is_30_bytes():
    data = open(os.path.join(FIX, "short.bin"), "rb").read()
    assert len(data) == 34
    c = ob.encode(data, wrapper=1)
    assert len(c) == 30
    assert zlib.decompress(c) == data


# src/zlib.rs:69-85
def test_zlib_header():
    for bits in (0 << 6, 1 << 6, 2 << 6, 3 << 6):
        h = ob.zlib_header(bits)
        assert (h[0] * 256 + h[1]) % 31 == 0
    assert ob.zlib_header(2 << 6) == b"\x78\x9c"
    assert ob.zlib_header(0) == b"\x78\x01"


def test_adler32_matches_zlib():
    for data in (b"", b"a", b"Wikipedia", bytes(range(256)) * 100, bytes([255]) * 70000):
        assert ob.adler32(data) == zlib.adler32(data)


# src/writer.rs:570-595 writer_sync: stream ends with 00 00 FF FF after flush()
def test_sync_marker():
    s = ob.Stream(ob.preset(ob.DEFAULT))
    s.write_all(b"some data to flush out" * 10)
    s.flush()
    assert s.output()[-4:] == b"\x00\x00\xff\xff"
    s.write_all(b"more")
    out = s.finish()
    assert inflate_raw(out) == b"some data to flush out" * 10 + b"more"


# src/lz77.rs:1101-1193 buffer_fill family: block length is exactly MAX_BUFFER_LENGTH values
def test_block_is_31744_values():
    import random
    rnd = random.Random(3)
    data = bytes(rnd.getrandbits(8) for _ in range(100000))
    c = ob.encode(data)
    blocks = ob.trace_blocks()
    assert blocks[0]["n_lz"] == 31744
    assert sum(b["in_bytes"] for b in blocks) == len(data)
    assert [b["bfinal"] for b in blocks] == [0] * (len(blocks) - 1) + [1]
    assert inflate_raw(c) == data


def test_empty_and_tiny():
    assert ob.encode(b"") == b"\x03\x00"
    for n in range(1, 6):
        d = bytes(range(n))
        assert inflate_raw(ob.encode(d)) == d
