"""Roundtrip/property tests of the CPU oracle over the reference's own fixtures
(tests/test.rs, src/lib.rs tests; SURVEY.md Appendix B group R).  CPU only."""
import glob
import os
import random
import zlib

import pytest

import oracle_binding as ob

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_inputs")
LEVELS = [ob.FAST, ob.DEFAULT, ob.BEST, ob.RLE, ob.HUFFMAN_ONLY]


def inflate_raw(b):
    d = zlib.decompressobj(-15)
    out = d.decompress(b) + d.flush()
    assert d.eof and d.unused_data == b""
    return out


def rd(name):
    return open(os.path.join(FIX, name), "rb").read()


# tests/test.rs:36-56,93-111 (high/fast/rle/default on pg11.txt)
@pytest.mark.parametrize("level", LEVELS)
def test_pg11_roundtrip(level):
    data = rd("pg11.txt")
    c = ob.encode(data, level=level)
    assert inflate_raw(c) == data
    assert len(c) < len(data)
    assert sum(b["in_bytes"] for b in ob.trace_blocks()) == len(data)


# tests/test.rs:68-76 issue_17, src/lib.rs:370-380, writer.rs tests: zeros of assorted sizes
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 258, 259, 260, 61000, 65535, 65536, 65537, 65794, 100000])
@pytest.mark.parametrize("level", LEVELS)
def test_zeros(n, level):
    data = bytes(n)
    assert inflate_raw(ob.encode(data, level=level)) == data


# tests/test.rs:138-145 issue_18_201911
@pytest.mark.parametrize("level", LEVELS)
def test_issue_18(level):
    data = rd("issue_18_201911.bin")
    assert inflate_raw(ob.encode(data, level=level)) == data


# tests/test.rs:147-161 afl_regressions_default_compression (+fast as in the fuzz target)
def test_afl_regressions():
    files = sorted(glob.glob(os.path.join(FIX, "afl", "*")))
    assert len(files) == 45
    for f in files + [os.path.join(FIX, "dump.bin")]:
        data = open(f, "rb").read()
        for level in (ob.DEFAULT, ob.FAST):
            c = ob.encode(data, level=level)
            assert inflate_raw(c) == data, f
            assert sum(b["in_bytes"] for b in ob.trace_blocks()) == len(data)


# tests/test.rs:78-91 issue_44 (26 214 400 bytes, 99.99 % zeros) -- #[ignore]d upstream, cheap here
def test_issue_44():
    data = zlib.decompress(rd("issue_44.zlib"))
    assert len(data) == 26214400
    c = ob.encode(data, level=ob.DEFAULT, wrapper=1)
    assert zlib.decompress(c) == data


# src/lib.rs:408-433 chunk_test: streaming output == one-shot output for any write chunking
@pytest.mark.parametrize("chunk", [1, 50, 400, 32768, 65794, 50000, 65794 + 258])
def test_chunked_write_equals_oneshot(chunk):
    data = rd("pg11.txt")
    if chunk == 1:
        data = data[:70000]
    for wrapper in (0, 1):
        opts = ob.preset(ob.DEFAULT, wrapper)
        one = ob.encode(data, opts=opts)
        s = ob.Stream(opts)
        for i in range(0, len(data), chunk):
            s.write_all(data[i:i + chunk])
        assert s.finish() == one


# src/lz77.rs:1081-1099 multiple_inputs; zlib_last_block src/lib.rs:370-380
def test_window_borders():
    cases = [bytes([22]) * 32768 + bytes([5, 2, 55, 11, 12]),
             bytes(32768 * 22) + b"tail",
             bytes(range(256)) * 128 + b"x" * 258,
             b"ab" * 40000]
    for data in cases:
        for level in LEVELS:
            assert inflate_raw(ob.encode(data, level=level)) == data
        assert zlib.decompress(ob.encode(data, wrapper=1)) == data


def test_random_and_mixed_seeded():
    rnd = random.Random(1234)
    words = [bytes(rnd.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rnd.randint(2, 9)))
             for _ in range(300)]
    for trial in range(12):
        kind = trial % 4
        n = rnd.choice([10, 1000, 40000, 70000, 140000])
        if kind == 0:
            data = bytes(rnd.getrandbits(8) for _ in range(n))
        elif kind == 1:
            data = b" ".join(rnd.choice(words) for _ in range(n // 5))[:n]
        elif kind == 2:
            data = bytes(rnd.choice([0, 0, 0, 1, 255]) for _ in range(n))
        else:
            base = bytes(rnd.getrandbits(8) for _ in range(997))
            data = (base * (n // 997 + 1))[:n]
        for level in LEVELS:
            c = ob.encode(data, level=level)
            assert inflate_raw(c) == data
            blocks = ob.trace_blocks()
            assert sum(b["in_bytes"] for b in blocks) == len(data)
            assert all(b["n_lz"] == 31744 for b in blocks[:-1])


# writer.rs:599-660 style: flush() mid-stream then continue; result still inflates
def test_sync_flush_midstream():
    data = rd("pg11.txt")
    s = ob.Stream(ob.preset(ob.DEFAULT))
    s.write_all(data[:70000])
    s.flush()
    mid = s.output()
    assert mid.endswith(b"\x00\x00\xff\xff")
    s.write_all(data[70000:])
    assert inflate_raw(s.finish()) == data
