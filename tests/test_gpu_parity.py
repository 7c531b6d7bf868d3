"""Parity tests proper: the HIP path (through the C ABI of include/mi355_deflate.h) against the
CPU oracle -- bit-exact, every level, the reference's own fixtures, edge sizes, quirks -- plus
size-independent properties at BASELINE sizes.  Needs a real MI355X: pytest -m gpu."""
import ctypes as C
import glob
import os
import sys
import zlib

import pytest
import torch  # noqa: F401  -- before the library: torch ships its own HIP runtime, and the copy that is loaded
#                              second in a process does not see the GPU (tests below pass tensors to the C ABI)

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd"))

import datagen
import oracle_binding as ob

pytestmark = pytest.mark.gpu

FIX = os.path.join(HERE, "golden", "ref_inputs")
LV = {"fast": (1, 0, 0), "default": (128, 32, 1), "best": (1768, 128, 1), "rle": (0, 0, 1),
      "huffman_only": (0, 0, 0)}


@pytest.fixture(scope="module")
def da():
    import deflate_amd
    return deflate_amd


@pytest.fixture(scope="module")
def ctx(da):
    c = da.Context(0)
    yield c
    c.close()


def inflate_raw(b):
    d = zlib.decompressobj(-15)
    out = d.decompress(b) + d.flush()
    assert d.eof and d.unused_data == b""
    return out


def agree(da, ctx, data, c, l, m, compat=1):
    ref = ob.encode(data, opts=ob.make_opts(c, l, m))
    rb = ob.trace_blocks()
    out = ctx.encode(data, da.CompressionOptions(c, l, m), compat=compat)
    if out != ref:
        bl = ctx.blocks()
        diff = next((i for i, (a, b) in enumerate(zip(bl, rb)) if a != b), None)
        raise AssertionError("HIP != oracle (%d vs %d bytes); first differing block %s: %s vs %s" % (
            len(out), len(ref), diff, bl[diff] if diff is not None else None,
            rb[diff] if diff is not None else None))
    assert ctx.blocks() == rb
    return ctx.info()


def test_library_is_the_hip_one(da, ctx):
    assert da.load().mi355_deflate_version() >= 100
    assert os.path.exists(da.LIB_PATH)


# tests/test.rs:36-56,93-111 + lib.rs:318-367: every level on pg11.txt -- here bit-exact, not just roundtrip
@pytest.mark.parametrize("level", list(LV))
def test_reference_fixtures_all_levels(da, ctx, level):
    for f in ["pg11.txt", "short.bin", "issue_18_201911.bin", "dump.bin"]:
        data = open(os.path.join(FIX, f), "rb").read()
        info = agree(da, ctx, data, *LV[level])
        assert info["in_len"] == len(data)


# tests/test.rs:147-161 afl_regressions_default_compression (+ fast, as the fuzz target does)
def test_afl_regressions(da, ctx):
    files = sorted(glob.glob(os.path.join(FIX, "afl", "*")))
    assert len(files) == 45
    for f in files:
        data = open(f, "rb").read()
        agree(da, ctx, data, *LV["default"])
        agree(da, ctx, data, *LV["fast"])


# tests/test.rs:58-63 block_type, lib.rs:382-391 deflate_short, compress.rs:333-345
def test_known_sizes(da, ctx):
    assert len(da.deflate_bytes_zlib(open(os.path.join(FIX, "short.bin"), "rb").read(), ctx)) == 30
    assert len(da.deflate_bytes(bytes([10, 10, 10, 10, 10, 55]), ctx)) == 5
    assert da.deflate_bytes(b"", ctx) == b"\x03\x00"


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 6, 258, 259, 260, 8191, 8192, 8193, 31743, 31744, 31745, 32767,
                               32768, 32769, 65535, 65536, 65537, 65794, 65795, 100000])
def test_edge_sizes(da, ctx, n):
    for level in LV:
        agree(da, ctx, bytes(n), *LV[level])
        agree(da, ctx, bytes(i & 0xFF for i in range(n)), *LV[level])


def test_random_block_fill_q1_q8(da, ctx):
    hit_q1 = 0
    for n in (31744, 31745, 63488, 63489, 95232, 100000, 300000):
        data = datagen.rng_bytes(n, n)
        for level in LV:
            hit_q1 += agree(da, ctx, data, *LV[level])["q1_rewarm"]
    assert hit_q1 > 0  # the hash re-warm quirk (lz77.rs:628-638) must have fired and been reproduced


def test_block_part_boundaries(da, ctx):
    """k_block_hist / k_pack cut a block's tokens into four parts of 7 936 (a part's first bit comes from the
    histograms of the parts before it): last blocks whose token count sits on, just before and just behind
    every part boundary.  huffman_only = one literal token per byte, so the byte count sets the token count."""
    pq = 31744 // 4
    base = datagen.text_like(31744 + 4 * pq + 8, 99)
    for k in range(0, 5):
        for d in (-1, 0, 1):
            n = 31744 + k * pq + d
            if n <= 0:
                continue
            agree(da, ctx, base[:n], *LV["huffman_only"])
            agree(da, ctx, datagen.rng_bytes(n, n), *LV["huffman_only"])
    for n in (1, pq - 1, pq, pq + 1, 2 * pq, 3 * pq + 1):
        agree(da, ctx, base[:n], *LV["huffman_only"])


def test_large_host_input_goes_over_in_pieces(da, ctx):
    """mi355_deflate_encode copies a host input of 16 MiB or more in two pieces (a head of about 0.3 n, then the rest) and starts sorting / walking
    the first epochs while the rest is on the bus (deflate_host.inc encode_host, launch_match_tables): same
    bytes as the oracle, for the sizes around a piece boundary, raw and zlib, and for a level on the other paths."""
    base = datagen.text_like(21_000_000, 123)
    for n in (16 << 20, (16 << 20) + 1, 20_000_003, 21_000_000):
        agree(da, ctx, base[:n], *LV["default"])
    assert da.deflate_bytes_zlib(base[:17_000_000], ctx) == ob.encode(base[:17_000_000], level=ob.DEFAULT, wrapper=1)
    agree(da, ctx, base[:17_000_000], *LV["fast"])
    agree(da, ctx, datagen.rng_bytes(17_000_000, 5), *LV["default"])  # Q1 fires: second pass over the head


@pytest.mark.parametrize("seed", range(6))
def test_mixed_inputs(da, ctx, seed):
    data = datagen.mixed([50000, 140000, 300000, 1000000, 70000, 2000000][seed], seed)
    for level in LV:
        if level == "best" and len(data) > 400000:
            data_l = data[:400000]
        else:
            data_l = data
        agree(da, ctx, data_l, *LV[level])


def test_text_like(da, ctx):
    data = datagen.text_like(4_000_000, 7)
    agree(da, ctx, data, *LV["default"])
    agree(da, ctx, data, *LV["fast"])
    agree(da, ctx, data[:500000], *LV["best"])


def test_custom_options(da, ctx):
    import random
    rnd = random.Random(42)
    data = datagen.mixed(200000, 99)
    for _ in range(12):
        c = rnd.choice([1, 2, 3, 7, 32, 128, 500, 4000])
        l = rnd.choice([3, 4, 8, 31, 32, 33, 64, 258, 1000, 40000])
        m = rnd.choice([0, 1])
        agree(da, ctx, data, c, l, m)


def test_small_one_shot_calls_between_parse_and_histograms(da, ctx):
    """One-shot calls of at most 2048 token segments (2 MiB) run everything between the speculative parse and the dense
    tokens -- the check of the segment chain, the repair of the entries that fail it, the scan of the token counts, the
    block table -- as ONE workgroup (k_small_fix), and pack their blocks with workgroups of 1024 threads: sizes on both
    sides of the limit, data with many full blocks (their last tokens are read for the Q1 / Q13 questions), data whose
    speculative entries need the repair, data whose speculation fails (the exact parse takes the same kernel), every level."""
    seg = 1024
    text = datagen.text_like(2100 * seg, 41)
    noise = datagen.rng_bytes(2100 * seg, 42)
    for n in (seg - 1, seg, seg + 1, 255 * seg + 7, 256 * seg, 256 * seg + 1, 700 * seg + 3, 1024 * seg - 1, 1024 * seg,
              1024 * seg + 1, 1500 * seg + 11, 2048 * seg - 1, 2048 * seg, 2048 * seg + 1):
        for level in LV:
            agree(da, ctx, text[:n], *LV[level])
            agree(da, ctx, noise[:n], *LV[level])
    # matches that cross block ends in the second and later windows (Q13), blocks that fill with a match
    rep = datagen.rng_bytes(40000, 43)
    for data in (rep * 20, (rep[:33000] + noise[:9000]) * 6, b"ab" * 300000, bytes(200 * seg),
                 datagen.mixed(900 * seg, 44), datagen.rng_bytes(300, 45) * 800):
        for level in LV:
            agree(da, ctx, data, *LV[level])
    # Periodic stretches of a few segments: inside one a run-up of 128 positions does not meet the true path, every boundary
    # in it fails the check, and the repair's wave parses on from the stretch's first segment (a fresh context: one whose
    # speculation failed a moment ago -- the zero fill above -- parses the exact way for a while)
    gaps = (lambda i: bytes(1500), lambda i: bytes(3000), lambda i: bytes([97 + i % 5, 98, 99, 100 + i % 3, 101]) * 600,
            lambda i: noise[i * 300:(i + 1) * 300] * 10)
    for g, gap in enumerate(gaps):
        data = b"".join(text[i * 7000:(i + 1) * 7000] + gap(i) for i in range(60))
        for level in ("default", "fast"):
            c2 = da.Context(0)
            try:
                info = agree(da, c2, data, *LV[level])
                assert info["spec_repaired"] > 0 and info["spec_fallback"] == 0, (g, level, info["spec_repaired"], info["spec_fallback"])
            finally:
                c2.close()
    # a sync flush behind everything that was written (the block table's sync marker), raw and zlib
    import io
    for n in (5000, 200 * seg + 5, 600 * seg):
        for wrapper, cls in ((0, da.DeflateEncoder), (1, da.ZlibEncoder)):
            _drive(lambda: cls(io.BytesIO(), da.CompressionOptions(*LV["default"]), ctx),
                   lambda: ob.Stream(ob.make_opts(*LV["default"], wrapper)), text[:n], [], flush_at_end=True)


def test_stage_clocks_are_optional(da):
    """mi355_deflate_info's per-stage clocks are events between the kernels of a call (5.7 us of idle queue each): a call runs
    without them unless MI355_CFG_STAGE_CLOCKS asks (1: every call, 2: calls of 32 MiB or more) -- stage_ms and match_ms then
    read 0, total_ms is the host's clock -- and the bytes are the same either way."""
    c = da.Context(0)
    try:
        data = open(os.path.join(FIX, "pg11.txt"), "rb").read()
        ref = ob.encode(data, level=ob.DEFAULT)
        seen = {}
        assert c.encode(data, da.Compression.Default) == ref  # (as a context comes: none)
        i = c.info()
        assert i["total_ms"] > 0 and i["match_ms"] == 0.0 and sum(i["stage_ms"].values()) == 0.0
        for mode in (2, 0, 1, 2):
            c.config(da.Context.CFG_STAGE_CLOCKS, mode)
            assert c.encode(data, da.Compression.Default) == ref
            i = c.info()
            assert i["total_ms"] > 0
            seen[mode] = (i["match_ms"], sum(i["stage_ms"].values()))
        assert seen[0] == (0.0, 0.0) and seen[2] == (0.0, 0.0)
        assert seen[1][0] > 0 and seen[1][1] > 0
        big = datagen.text_like(40 << 20, 77)  # (mode 2: from 32 MiB on)
        c.config(da.Context.CFG_STAGE_CLOCKS, 2)
        out = _encode_resident(da, c, big, da.Compression.Fast)  # (a host call of this size is worked on in pieces, which have clocks of their own)
        assert c.info()["match_ms"] > 0
        assert inflate_raw(out) == big
        with pytest.raises(da.DeflateError):
            c.config(da.Context.CFG_STAGE_CLOCKS, 3)
    finally:
        c.close()


def test_periodic_and_runs(da, ctx):
    for data in (b"ab" * 70000, b"abc" * 50000, bytes(100) + b"x" + bytes(200000),
                 datagen.rng_bytes(300, 5) * 500, datagen.rng_bytes(32768, 6) * 4,
                 datagen.rng_bytes(32769, 7) * 3, datagen.rng_bytes(32767, 8) * 3):
        for level in LV:
            agree(da, ctx, data, *LV[level])


def test_unsupported_and_errors(da, ctx):
    with pytest.raises(da.DeflateError) as e:
        ctx.encode(b"abc" * 100, da.CompressionOptions(16, 2, da.MatchingType.Lazy))
    assert e.value.code == da.E_UNSUPPORTED


# lib.rs:446-485 zlib roundtrips + checksum.rs: zlib framing and Adler-32 computed on the GPU
def test_zlib_wrapper(da, ctx):
    for data in (b"", b"a", open(os.path.join(FIX, "pg11.txt"), "rb").read(), datagen.rng_bytes(70000, 3),
                 bytes(70000)):
        z = da.deflate_bytes_zlib(data, ctx)
        assert z == ob.encode(data, level=ob.DEFAULT, wrapper=1)
        assert zlib.decompress(z) == data


# writer.rs tests: ZlibEncoder / DeflateEncoder write_all + finish; lib.rs:408-433 chunk invariance
@pytest.mark.parametrize("chunk", [1, 50, 400, 32768, 65794, 50000])
def test_streaming_encoders(da, ctx, chunk):
    import io
    data = open(os.path.join(FIX, "pg11.txt"), "rb").read()
    if chunk == 1:
        data = data[:20000]
    for cls, wrapper in ((da.DeflateEncoder, 0), (da.ZlibEncoder, 1)):
        enc = cls(io.BytesIO(), da.Compression.Default, ctx)
        for i in range(0, len(data), chunk):
            enc.write_all(data[i:i + chunk])
        if wrapper:
            assert enc.checksum() == zlib.adler32(data)
        out = enc.finish().getvalue()
        assert out == ob.encode(data, level=ob.DEFAULT, wrapper=wrapper)


# bug-for-bug with MI355_COMPAT_Q13, a valid stream without
def test_q13_modes(da, ctx):
    from test_stages_vs_oracle import q13_case
    hit = 0
    for seed in range(1, 8):
        d = q13_case(seed)
        if d is None:
            continue
        ref = ob.encode(d, level=ob.DEFAULT)
        if not ob.last_hazards():
            continue
        hit += 1
        assert ctx.encode(d, da.Compression.Default, compat=da.COMPAT_Q13) == ref
        assert ctx.info()["q13_hits"] >= 1
        sane = ctx.encode(d, da.Compression.Default, compat=0)
        assert inflate_raw(sane) == d and len(sane) == len(ref)
        if hit >= 2:
            break
    assert hit >= 1


# device-resident API through torch tensors (no host copies in the path)
def test_device_api(da, ctx):
    import torch
    data = datagen.text_like(3_000_000, 11)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    cap = da.bound(len(data)) + 8
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    n = ctx.encode_device(t.data_ptr(), len(data), out.data_ptr(), cap, da.Compression.Default,
                          stream=torch.cuda.current_stream().cuda_stream)
    got = bytes(out[:n].cpu().numpy())
    assert got == ob.encode(data, level=ob.DEFAULT)
    assert ctx.adler32_device(t.data_ptr(), len(data)) == zlib.adler32(data)
    # unaligned device input pointer takes the byte-load path
    n2 = ctx.encode_device(t.data_ptr() + 1, len(data) - 1, out.data_ptr(), cap, da.Compression.Default)
    assert bytes(out[:n2].cpu().numpy()) == ob.encode(data[1:], level=ob.DEFAULT)


# BASELINE config 2: 256 MiB zero fill, RLE path; oracle is fast on zeros so this stays bit-exact
def test_config2_zero_fill_256mib(da, ctx):
    import torch
    n = 256 * 1024 * 1024
    t = torch.zeros(n, dtype=torch.uint8, device="cuda")
    cap = da.bound(n) + 8
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    k = ctx.encode_device(t.data_ptr(), n, out.data_ptr(), cap, da.CompressionOptions.rle())
    got = bytes(out[:k].cpu().numpy())
    info = ctx.info()
    assert info["n_blocks"] == 33  # 1 literal + 1 040 448 matches -> 33 blocks of 31744 values
    ref = ob.encode(bytes(n), level=ob.RLE)
    assert got == ref
    d = zlib.decompressobj(-15)
    total = 0
    buf = got
    while buf:
        piece = d.decompress(buf, 1 << 24)
        assert piece.count(0) == len(piece)
        total += len(piece)
        buf = d.unconsumed_tail
    assert total == n


# BASELINE config 3 size (100 MB text, Default): size-independent properties -- inflate round trip,
# every non-final block holds exactly 31744 tokens, sum of block bytes == input, plus bit-exactness of a
# prefix-sized sample against the oracle
def test_config3_full_size_properties(da, ctx):
    import torch
    n = 100_000_000
    data = datagen.text_like(n, 0x656E77696B38)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    cap = da.bound(n) + 8
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    k = ctx.encode_device(t.data_ptr(), n, out.data_ptr(), cap, da.Compression.Default)
    got = bytes(out[:k].cpu().numpy())
    bl = ctx.blocks()
    assert all(b["n_lz"] == 31744 for b in bl[:-1])
    assert sum(b["in_bytes"] for b in bl) == n
    assert [b["bfinal"] for b in bl] == [0] * (len(bl) - 1) + [1]
    assert zlib.crc32(inflate_raw(got)) == zlib.crc32(data)
    # the stream for a 16 MB prefix is a different stream (BFINAL position) but must equal the oracle
    m = 16_000_000
    k2 = ctx.encode_device(t.data_ptr(), m, out.data_ptr(), cap, da.Compression.Default)
    assert bytes(out[:k2].cpu().numpy()) == ob.encode(data[:m], level=ob.DEFAULT)


def _big_gold(name):
    import json
    return json.load(open(os.path.join(HERE, "golden", "big_digests.json")))["digests"][name]


def _encode_resident(da, ctx, data, options):
    import torch
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    cap = da.bound(len(data)) + 8
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    k = ctx.encode_device(t.data_ptr(), len(data), out.data_ptr(), cap, options)
    return bytes(out[:k].cpu().numpy())


# BASELINE config 3 at full size against the oracle's stream, by its committed digest
# (tests/golden/big_digests.json, written by gen_big_digests.py and re-derived by the CPU suite)
def test_config3_full_size_equals_the_oracle(da, ctx):
    import hashlib
    g = _big_gold("config3_enwik8_like_100MB_default")
    data = datagen.text_like(100_000_000, 0x656E77696B38)
    assert hashlib.sha256(data).hexdigest() == g["in_sha256"]
    got = _encode_resident(da, ctx, data, da.Compression.Default)
    assert [len(got), hashlib.sha256(got).hexdigest()] == [g["out_len"], g["out_sha256"]]


# BASELINE config 4: the Silesia-like mix (212 100 000 bytes: text, binary records, database rows, 16-bit
# samples, noise) at Compression::Best -- compression_options.rs:126-133, quarter budget lz77.rs:351-355 --
# byte for byte against the oracle (digest), plus the block invariants and an inflate round trip
def test_config4_silesia_like_best_full_size(da, ctx):
    import hashlib
    g = _big_gold("config4_silesia_like_best")
    data = datagen.silesia_like(0x53494C45)
    assert len(data) == g["in_len"] and hashlib.sha256(data).hexdigest() == g["in_sha256"]
    got = _encode_resident(da, ctx, data, da.Compression.Best)
    bl = ctx.blocks()
    assert sum(b["in_bytes"] for b in bl) == len(data)
    assert all(b["n_lz"] == 31744 for b in bl[:-1])
    assert {b["btype"] for b in bl} == {0, 2} or {b["btype"] for b in bl} == {0, 1, 2}  # stored (noise) and dynamic
    assert [len(got), hashlib.sha256(got).hexdigest()] == [g["out_len"], g["out_sha256"]]
    assert zlib.crc32(inflate_raw(got)) == zlib.crc32(data)


# BASELINE config 5 workload: the web-text input (generated per 1 MiB segment from seed ^ index),
# its first 256 MiB sharded stream-exact over eight virtual ranks on this one GPU -- the exchanges of the
# distributed driver, byte for byte the oracle's single stream
def test_config5_webtext_sharded_over_8_virtual_ranks(da, ctx):
    import hashlib
    import shard
    g = _big_gold("config5_webtext_256MiB_default")
    data = datagen.webtext(256 << 20)
    assert hashlib.sha256(data).hexdigest() == g["in_sha256"]
    ctxs = [da.Context(0) for _ in range(8)]
    try:
        got = shard.encode_p1_virtual(da, ctxs, data, da.Compression.Default, compat=1)
    finally:
        for c in ctxs:
            c.close()
    assert [len(got), hashlib.sha256(got).hexdigest()] == [g["out_len"], g["out_sha256"]]
    one = _encode_resident(da, ctx, data, da.Compression.Default)
    assert one == got


# BASELINE config 5 at its stated size: 8 GiB of the web-text input, against the oracle's digests of the whole stream
# (tests/golden/config5_digest.json, gen_config5_digest.py: six minutes of oracle).  The input is generated on the box's cores a
# MiB segment at a time straight into device memory; (a) one GPU walks it in 512 MiB ranges (mi355_deflate_encode_device), raw and
# zlib; (b) mi355_deflate_encode_multi_device cuts it over eight ranks of 1 GiB -- here all on this one device.
def test_config5_at_8_gib(da):
    import hashlib
    import json
    import multiprocessing as mp
    import torch
    gold = json.load(open(os.path.join(HERE, "golden", "config5_digest.json")))["digests"]
    N = gold["raw"]["in_len"]
    assert N == 8 << 30
    d_in = torch.empty(N + 64, dtype=torch.uint8, device="cuda")
    d_in[N:] = 0
    hin = hashlib.sha256()
    per = 64
    with mp.get_context("spawn").Pool(min(48, max(2, (os.cpu_count() or 4) - 2))) as pool:
        pending = []
        nxt, done, n_seg = 0, 0, N // datagen.WEB_SEGMENT
        while done < N:
            while len(pending) < 4 and nxt < n_seg:
                pending.append(pool.map_async(datagen.webtext_segment_bytes, range(nxt, min(nxt + per, n_seg)), chunksize=2))
                nxt = min(nxt + per, n_seg)
            b = b"".join(pending.pop(0).get())
            hin.update(b)
            d_in[done:done + len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
            done += len(b)
    assert hin.hexdigest() == gold["raw"]["in_sha256"]

    def digest_of(d_out, n):
        h = hashlib.sha256()
        for i in range(0, n, 256 << 20):
            h.update(bytes(d_out[i:min(n, i + (256 << 20))].cpu().numpy()))
        return [n, h.hexdigest()]
    cap = da.bound(N) + 64
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    ctx = da.Context(0)
    try:
        for name, wrapper in (("raw", 0), ("zlib", 1)):
            n = ctx.encode_device(d_in.data_ptr(), N, d_out.data_ptr(), cap, da.Compression.Default, wrapper=wrapper)
            assert ctx.info()["passes"] >= 16, "8 GiB are sixteen ranges of 512 MiB"
            assert digest_of(d_out, n) == [gold[name]["out_len"], gold[name]["out_sha256"]], name
    finally:
        ctx.close()
    m = da.MultiGpu([0] * 8)
    try:
        lay = [m.layout(N, r) for r in range(8)]
        assert lay[0]["n_ranks"] == 8 and all(L["hi"] - L["lo"] == 1 << 30 for L in lay)
        d_out.fill_(0x33)
        n = m.encode_device([d_in.data_ptr() + L["g_lo"] for L in lay], N, d_out.data_ptr(), cap, da.Compression.Default)
        assert digest_of(d_out, n) == [gold["raw"]["out_len"], gold["raw"]["out_sha256"]], "eight ranks"
    finally:
        m.close()


# MI355_FLUSH_SYNC == fresh reference encoder: write_all(chunk); flush() (writer.rs:134-137,
# compress.rs:256-261) -- the chunk form the multi-GPU stitch concatenates
def test_sync_flush_chunks_and_stitch(da, ctx):
    data = datagen.text_like(1_500_000, 21)
    cuts = [0, 400000, 400001, 1_000_000, len(data)]
    pieces = []
    for i in range(len(cuts) - 1):
        chunk = data[cuts[i]:cuts[i + 1]]
        last = i == len(cuts) - 2
        got = ctx.encode(chunk, da.Compression.Default, flush=da.FLUSH_FINISH if last else da.FLUSH_SYNC)
        s = ob.Stream(ob.preset(ob.DEFAULT))
        s.write_all(chunk)
        if last:
            exp = s.finish()
        else:
            s.flush()
            exp = s.output()
            assert exp.endswith(b"\x00\x00\xff\xff")
        assert got == exp
        pieces.append(got)
    assert inflate_raw(b"".join(pieces)) == data
    # zlib header on a sync chunk, no trailer
    z = ctx.encode(data[:50000], da.Compression.Default, wrapper=1, flush=da.FLUSH_SYNC)
    s = ob.Stream(ob.preset(ob.DEFAULT, 1))
    s.write_all(data[:50000])
    s.flush()
    assert z == s.output()


# Stream-exact (P1) sharding: several virtual ranks on this one GPU run the sharded phases and exchange
# exit tables / token counts / straddling tokens / block costs exactly as the distributed driver does;
# the stitched stream must equal the oracle's stream for the WHOLE input, byte for byte.
@pytest.mark.parametrize("world", [2, 3, 5])
def test_p1_sharded_stream_exact(da, world):
    import shard
    ctxs = [da.Context(0) for _ in range(world)]
    try:
        cases = [("text", datagen.text_like(3_000_000, 31), "default"),
                 ("mixed", datagen.mixed(2_500_000, 8), "default"),
                 ("random", datagen.rng_bytes(1_200_000, 9), "default"),   # stored blocks, Q1 on rank 0
                 ("zeros", bytes(40_000_000), "default"),                 # never-merging 258-byte steps
                 ("text-fast", datagen.text_like(2_000_000, 32), "fast"),
                 ("text-rle", datagen.text_like(1_500_000, 33), "rle")]
        for name, data, level in cases:
            c, l, m = LV[level]
            ref = ob.encode(data, opts=ob.make_opts(c, l, m))
            got = shard.encode_p1_virtual(da, ctxs, data, da.CompressionOptions(c, l, m), compat=1)
            assert got == ref, "%s/%s world=%d: sharded stream differs (%d vs %d bytes)" % (
                name, level, world, len(got), len(ref))
    finally:
        for c in ctxs:
            c.close()


def test_p1_sharded_ranges_enter_by_speculation(da):
    """A rank's range finds its entry like a segment does -- a run-up into its history -- and the driver checks that every
    rank was entered where the rank before it was left (shard.p1_spec_entries): text takes that way and no exit table is
    made; zero fill (a 258-byte match after the other: paths never meet) does not, and the exit tables decide.  Either
    way the entries are those of the exact way."""
    import shard
    import torch
    world = 4
    ctxs = [da.Context(0) for _ in range(world)]
    try:
        for name, data, want_chain in (("text", datagen.text_like(6_000_000, 0x51), True),
                                       ("mixed", datagen.mixed(5_000_000, 0x52), None),   # (its periodic pieces: either way)
                                       ("zeros", bytes(5_000_000), False)):
            lay = [shard.p1_layout(len(data), r, world) for r in range(world)]
            bufs, shards = [], []
            for r in range(world):
                L = lay[r]
                t = torch.frombuffer(bytearray(data[L["g_lo"]:L["g_hi"]]) + bytearray(16), dtype=torch.uint8).to("cuda:0")
                bufs.append(t)
                shards.append(da.Shard(ctxs[r], t.data_ptr(), L["g_hi"] - L["g_lo"], L["lo"], L["hi"], L["g_lo"], len(data),
                                       da.Compression.Default, 1))
            try:
                specs = [s.spec() for s in shards]
                fast = shard.p1_spec_entries(lay, specs)
                exact = shard.p1_entries(lay, [s.exit_table() for s in shards])
                assert want_chain is None or (fast is not None) == want_chain, (name, specs)
                if fast is not None:
                    assert fast == exact, name
                    # the tokens are there: emit at the speculation's entry hands them out; emit anywhere else takes the exact
                    # way (exit tables, way down, k_emit<0>) -- and coming back to the true entry afterwards does so too
                    for r in range(world):
                        n1, p1 = shards[r].emit(fast[r] - lay[r]["g_lo"])
                        spec_tok = torch.empty(n1, dtype=torch.int32, device="cuda:0")
                        shard.ctypes_copy_d2d(spec_tok.data_ptr(), p1, 4 * n1)
                        shards[r].emit(fast[r] - lay[r]["g_lo"] + 1 if r else 1)
                        assert not shards[r].spec()[0]
                        n2, p2 = shards[r].emit(fast[r] - lay[r]["g_lo"])
                        exact_tok = torch.empty(n2, dtype=torch.int32, device="cuda:0")
                        shard.ctypes_copy_d2d(exact_tok.data_ptr(), p2, 4 * n2)
                        assert n1 == n2 and torch.equal(spec_tok, exact_tok), (name, r)
            finally:
                for s in shards:
                    s.close()
            got = shard.encode_p1_virtual(da, ctxs, data, da.Compression.Default, compat=1)
            c, l, m = LV["default"]
            assert got == ob.encode(data, opts=ob.make_opts(c, l, m)), name
    finally:
        for c in ctxs:
            c.close()


def test_p1_sharded_ranges_shorter_than_a_block(da):
    """Ranges that yield fewer tokens than it takes to reach the next block boundary own no block: their tokens
    go to a block that began several ranks to the left, and that block's owner takes its tail from the heads of
    several ranks (shard.p1_token_plan, mi355_shard_blocks_ex).  Round 1 refused such splits."""
    import shard
    world = 8
    ctxs = [da.Context(0) for _ in range(world)]
    try:
        # (a rank's range is a multiple of 32 KiB: shard.shard_range)
        cases = [("zeros", bytes(3_000_000), "default"),                      # ~1 400 tokens per rank
                 ("zeros-small", bytes(300_000), "default"),                  # 127 tokens per rank
                 ("text-small", datagen.text_like(300_000, 41), "default"),   # ~10 000 tokens per rank
                 ("text-best", datagen.text_like(300_000, 42), "best"),
                 ("period", (datagen.rng_bytes(300, 3) * 4000)[:1_000_000], "fast"),
                 ("mixed", datagen.mixed(400_000, 43), "default")]
        for name, data, level in cases:
            c, l, m = LV[level]
            ref = ob.encode(data, opts=ob.make_opts(c, l, m))
            got = shard.encode_p1_virtual(da, ctxs, data, da.CompressionOptions(c, l, m), compat=1)
            assert got == ref, "%s/%s: sharded stream differs (%d vs %d bytes)" % (name, level, len(got), len(ref))
    finally:
        for c in ctxs:
            c.close()


# The parse finds every segment's entry by speculation -- a run-up of 128 positions in front of the segment, paths merge
# for good once they share a restart position -- and CHECKS the chain of entries and exits (k_emit<true>, k_scan_a); long
# periodic data (zero fill: a 258-byte match after the other, a path never meets the one that started a byte later) fails
# the check, the call is parsed again with exit tables and the table tree, and the context parses that way for a while.
# Same bytes either way.
def test_speculative_segment_entries_and_their_fallback(da):
    c = da.Context(0)
    try:
        text = datagen.text_like(5_000_000, 0xE1)
        zeros = bytes(3_000_001)
        per = (datagen.rng_bytes(301, 7) * 20000)[:4_000_000]
        for lv in ("default", "fast", "best", "rle"):
            a, l, m = LV[lv]
            o = da.CompressionOptions(a, l, m)
            c2 = da.Context(0)
            try:
                assert c2.encode(text, o) == ob.encode(text, opts=ob.make_opts(a, l, m, 0))
                if lv != "rle":  # (the run level sees text as literals: every position a restart position)
                    assert c2.info()["spec_fallback"] == 0, lv
                assert c2.encode(zeros, o) == ob.encode(zeros, opts=ob.make_opts(a, l, m, 0))
                assert c2.info()["spec_fallback"] >= 1, lv  # (the number of segment boundaries that failed the check)
                for _ in range(3):  # the exact parse for a while: no second attempt, no fallback
                    assert c2.encode(per, o) == ob.encode(per, opts=ob.make_opts(a, l, m, 0))
                    assert c2.info()["spec_fallback"] == 0
            finally:
                c2.close()
        # mixed data with the seams of its pieces, every level, speculation on (a fresh context per case)
        mix = datagen.mixed(6_000_000, 0xE2) + zeros[:700_000] + text[:900_000] + per[:500_000]
        for lv in ("default", "fast", "best", "huffman_only"):
            a, l, m = LV[lv]
            c3 = da.Context(0)
            try:
                assert c3.encode(mix, da.CompressionOptions(a, l, m)) == ob.encode(mix, opts=ob.make_opts(a, l, m, 0)), lv
            finally:
                c3.close()
    finally:
        c.close()


# ---- SURVEY section 8 row h: ONE input over N devices in ONE call of the C ABI (mi355_deflate_encode_multi) ----
# One process, a context and a host thread per rank; here the ranks share device 0 (what a one-GPU box can run: the
# phases, the exchanges through host memory, the seam words and the framing are the code an 8-GPU node runs, only the
# kernels of the ranks queue on one device and the peer copy is a device-to-device copy).
@pytest.mark.parametrize("world", [2, 3, 5, 8])
def test_multi_gpu_encode_in_one_call(da, world):
    m = da.MultiGpu([0] * world)
    try:
        cases = [("text", datagen.text_like(14_000_000, 0xC1), ("default", "fast", "best")),
                 ("mixed", datagen.mixed(9_500_000, 0xC2), ("default",)),
                 ("noise", datagen.rng_bytes(8_388_608 + 12345, 0xC3), ("default",)),   # stored blocks across every seam, Q1
                 ("zeros", bytes(12_000_000), ("default", "rle")),                      # ranges shorter than a block
                 ("period", (datagen.rng_bytes(300, 3) * 40000)[:9_000_001], ("fast", "huffman_only")),
                 ("short", datagen.text_like(1_500_000, 0xC4), ("default",)),            # fewer ranks than devices
                 ("tiny", b"hello hello hello", ("default",)), ("empty", b"", ("default",))]
        for name, data, levels in cases:
            for lv in levels:
                c, l, mt = LV[lv]
                o = da.CompressionOptions(c, l, mt)
                want = ob.encode(data, opts=ob.make_opts(c, l, mt, 0))
                got = m.encode(data, o)
                assert got == want, "%s/%s over %d ranks: %d vs %d bytes" % (name, lv, world, len(got), len(want))
        # zlib and gzip framing: every rank sums its own range, the sums are folded in rank order
        data = datagen.text_like(11_000_000, 0xC5) + datagen.rng_bytes(700_000, 0xC6)
        c, l, mt = LV["default"]
        assert m.encode(data, da.Compression.Default, wrapper=1) == ob.encode(data, opts=ob.make_opts(c, l, mt, 1))
        assert zlib.decompress(m.encode(data, da.Compression.Default, wrapper=1)) == data
        hdr = da.BLANK_GZIP_HEADER
        assert m.encode(data, da.Compression.Default, wrapper=2) == ob.encode_gzip(data, hdr, opts=ob.make_opts(c, l, mt, 0))
        hdr2 = bytes([0x1f, 0x8b, 8, 8, 1, 2, 3, 4, 0, 3]) + b"name.txt\0"
        assert m.encode(data, da.Compression.Default, wrapper=2, gzip_header=hdr2) == ob.encode_gzip(
            data, hdr2, opts=ob.make_opts(c, l, mt, 0))
        lay = [m.layout(len(data), r) for r in range(world)]
        assert lay[0]["n_ranks"] == world and lay[0]["lo"] == 0 and lay[-1]["hi"] == len(data)
        assert all(lay[r]["hi"] == lay[r + 1]["lo"] and lay[r + 1]["lo"] % 32768 == 0 for r in range(world - 1))
    finally:
        m.close()


def test_multi_gpu_more_ranks_than_devices(da):
    """An input beyond what one rank takes per device (2 x MI355_CFG_RANGE_BYTES of rank 0's context) is cut into MORE ranks
    than devices -- rank r on device r % n, the ranks of a device one after the other in every phase: no size ceiling short of
    the devices' memory (round-4 review: 8 GiB over two devices was MI355_E_UNSUPPORTED).  With 16 MiB ranges: two devices,
    four ranks each, against the oracle -- host buffers and resident shards, raw / zlib / gzip, and the handle serialises
    callers (two threads on one handle)."""
    import threading
    import torch
    m = da.MultiGpu([0, 0])
    try:
        m.config(da.Context.CFG_RANGE_BYTES, 16 << 20)
        data = datagen.text_like(150_000_000, 0xE1)[:-7] + datagen.rng_bytes(3_000_000, 0xE2) + bytes(9_000_000) + datagen.mixed(40_000_000, 0xE3)
        lay = [m.layout(len(data), r) for r in range(8)]
        assert lay[0]["n_ranks"] == 8 and lay[0]["lo"] == 0 and lay[7]["hi"] == len(data)
        assert all(lay[r]["hi"] - lay[r]["lo"] <= (32 << 20) + 8 * 32768 for r in range(8))
        c, l, mt = LV["default"]
        for wrapper in (0, 1, 2):
            want = ob.encode(data, opts=ob.make_opts(c, l, mt, wrapper)) if wrapper < 2 else ob.encode_gzip(
                data, da.BLANK_GZIP_HEADER, opts=ob.make_opts(c, l, mt, 0))
            got = m.encode(data, da.Compression.Default, wrapper=wrapper)
            assert got == want, (wrapper, len(got), len(want))
        want = ob.encode(data, opts=ob.make_opts(c, l, mt, 0))
        bufs = [torch.frombuffer(bytearray(data[L["g_lo"]:L["g_hi"]]) + bytearray(64), dtype=torch.uint8).cuda() for L in lay]
        cap = da.bound(len(data)) + 64
        d_out = torch.full((cap,), 0x55, dtype=torch.uint8, device="cuda")
        n = m.encode_device([b.data_ptr() for b in bufs], len(data), d_out.data_ptr(), cap, da.Compression.Default)
        assert bytes(d_out[:n].cpu().numpy()) == want
        # a level with few tokens per range (ranges shorter than a block: tails made of several ranks' heads)
        z = bytes(70_000_000)
        assert m.encode(z, da.CompressionOptions(*LV["rle"])) == ob.encode(z, opts=ob.make_opts(*LV["rle"], 0))
        # two callers on one handle: the calls follow each other
        small = data[:40_000_000]
        want_small = ob.encode(small, opts=ob.make_opts(c, l, mt, 1))
        res = [None, None]

        def call(i):
            res[i] = m.encode(small, da.Compression.Default, wrapper=1)
        th = [threading.Thread(target=call, args=(i,)) for i in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert res[0] == want_small and res[1] == want_small
        assert torch.cuda.current_device() == 0
    finally:
        m.close()


@pytest.mark.parametrize("stitch", [0, 1])
def test_multi_gpu_encode_device_resident(da, stitch):
    """The same call with every rank's bytes already on its device and the stream assembled in rank 0's device memory:
    the packed ranges arrive by peer copies (stitch 0) or by ncclSend / ncclRecv (MI355_CFG_MULTI_STITCH = 1: RCCL found at
    run time; with the ranks on one device every pair is a send to and a receive from the communicator's own rank), the seam
    words by one small kernel."""
    import torch
    for world in (2, 4):
        m = da.MultiGpu([0] * world)
        try:
            m.config(da.Context.CFG_MULTI_STITCH, stitch)
            for data, lv, wrapper in ((datagen.text_like(9_000_000, 0xD1), "default", 0), (datagen.mixed(7_000_000, 0xD2), "best", 1),
                                      (bytes(10_000_000), "default", 2), (datagen.rng_bytes(5_000_000, 0xD3), "default", 0)):
                c, l, mt = LV[lv]
                bufs = []
                for r in range(world):
                    L = m.layout(len(data), r)
                    bufs.append(torch.frombuffer(bytearray(data[L["g_lo"]:L["g_hi"]]) + bytearray(64), dtype=torch.uint8).cuda())
                cap = da.bound(len(data)) + 64
                d_out = torch.full((cap,), 0xAA, dtype=torch.uint8, device="cuda")  # (nothing relies on a cleared buffer)
                n = m.encode_device([b.data_ptr() for b in bufs], len(data), d_out.data_ptr(), cap, da.CompressionOptions(c, l, mt),
                                    wrapper=wrapper)
                want = ob.encode(data, opts=ob.make_opts(c, l, mt, wrapper)) if wrapper < 2 else ob.encode_gzip(
                    data, da.BLANK_GZIP_HEADER, opts=ob.make_opts(c, l, mt, 0))
                assert bytes(d_out[:n].cpu().numpy()) == want, (world, lv, wrapper)
                t = m.trace()
                assert t["call"] > 0
        finally:
            m.close()


def _p1_dist_worker(rank, world, port, q, backend="gloo"):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd"))
    sys.path.insert(0, HERE)
    import datagen as dg
    import deflate_amd as da
    import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    ctx = da.Context(0)
    ok = []
    # text: every rank's range is entered by speculation where the rank before it was left (round 0 of the driver decides);
    # zero fill in front of text: the chain does not hold and the exit tables are exchanged (round 1)
    for data in (dg.text_like(6_000_000, 77), bytes(3_500_000) + dg.text_like(2_500_000, 78)):
        total = len(data)  # every rank regenerates the same input and keeps its slice
        L = shard.p1_layout(total, rank, world)
        d_ext = torch.frombuffer(bytearray(data[L["g_lo"]:L["g_hi"]]) + bytearray(16), dtype=torch.uint8).cuda()
        for wrapper in (0, 1, 2):  # raw; zlib and gzip: every rank sums its own range, rank 0 folds and frames
            out, n = shard.encode_p1_dist(da, ctx, d_ext, L, total, rank, world, da.Compression.Default, compat=1,
                                          comm_device="cpu" if backend == "gloo" else None, wrapper=wrapper)
            if rank == 0:
                import oracle_binding as ob2
                ref = (ob2.encode_gzip(data, da.BLANK_GZIP_HEADER, level=ob2.DEFAULT) if wrapper == 2
                       else ob2.encode(data, level=ob2.DEFAULT, wrapper=wrapper))
                ok.append((wrapper, bytes(out.cpu().numpy()) == ref, n, len(ref)))
    if rank == 0:
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


# the distributed driver itself (torch.distributed, gloo here; RCCL on a multi-GPU node), three processes
# sharing this one GPU
def test_p1_distributed_driver_gloo():
    import torch.multiprocessing as mp
    world = 3
    mctx = mp.get_context("spawn")
    q = mctx.Queue()
    port = 29600 + os.getpid() % 1000
    procs = [mctx.Process(target=_p1_dist_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert [x[:2] for x in ok] == [(0, True), (1, True), (2, True)] * 2 and all(x[2] == x[3] for x in ok), ok


# The same driver over the backend a multi-GPU node uses -- "nccl" (= RCCL) with the exchanged tensors on the device -- as far
# as ONE GPU can take it: a world of one rank.  Every collective of the driver (all_gather_into_tensor of int64 / int32 / uint8
# device tensors) goes through RCCL; the point-to-point stitch has no peer here and is covered by the gloo run above.
def test_p1_distributed_driver_rccl_world_of_one():
    import torch.multiprocessing as mp
    mctx = mp.get_context("spawn")
    q = mctx.Queue()
    port = 29700 + os.getpid() % 1000
    p = mctx.Process(target=_p1_dist_worker, args=(0, 1, port, q, "nccl"))
    p.start()
    ok = q.get(timeout=300)
    p.join(120)
    assert p.exitcode == 0
    assert [x[:2] for x in ok] == [(0, True), (1, True), (2, True)] * 2 and all(x[2] == x[3] for x in ok), ok


# flush() = Flush::Sync with window retention (writer.rs:134-137, 571-660; tests/test.rs:113-123
# issue_26 "write after flush"): the GPU stream must equal the reference encoder driven with the same
# write/flush sequence.
def _drive(enc_new, ref_new, data, cuts, flush_at_end=False, chunk=0):
    import io
    enc = enc_new()
    ref = ref_new()

    def put(piece):
        # one write() call per `chunk` bytes (the reference's hash re-warm after a flush depends on the
        # call pattern, lz77.rs:601-638)
        step = chunk or max(len(piece), 1)
        i = 0
        while i < len(piece):
            j = i + step
            enc.write_all(piece[i:j])
            ref.write_all(piece[i:j])
            i = j

    prev = 0
    for c in cuts:
        put(data[prev:c])
        enc.flush()
        ref.flush()
        prev = c
    put(data[prev:])
    if flush_at_end:
        enc.flush()
        ref.flush()
    got = enc.finish().getvalue()
    exp = ref.finish()
    assert got == exp, "stream with flush points %s differs (%d vs %d bytes)" % (cuts, len(got), len(exp))
    return got


@pytest.mark.parametrize("level", ["default", "fast", "rle", "best"])
def test_flush_with_window_retention(da, ctx, level):
    import io
    import random
    c, l, m = LV[level]
    rnd = random.Random(5)
    texts = [datagen.text_like(260000 if level != "best" else 120000, 41), datagen.mixed(200000, 12),
             datagen.rng_bytes(90000, 13), bytes(150000),
             # periodic data: every position is the nearest candidate of a later one, so a position that is
             # missing from (or misfiled in) the hash chains shows up as a different distance
             (datagen.rng_bytes(300, 3) * 700)[:200000 if level != "best" else 90000],
             (datagen.rng_bytes(4099, 4) * 40)[:150000 if level != "best" else 90000]]
    for data in texts:
        n = len(data)
        cut_sets = [([n // 3], 0), ([1000, 70000], 0), ([0], 0), ([5, 5, 40000], 0), ([n - 3], 0),
                    ([32768, 65536, 65537 + 2], 0), (sorted(rnd.sample(range(3, n - 3), 6)), 0),
                    # flushes inside the first window followed by small writes: hash re-warm at the flush point
                    ([5, 40000], 0), ([100, 200, 300, 50000], 0), ([32768, 40000], 0), ([32767, 33000, 70000], 0),
                    ([31744], 0), ([31744, 63488], 0), ([100], 1500), ([3, 9], 7000), ([31000], 33000), ([31000], 40000), (sorted(rnd.sample(range(3, 32768), 5)), 1500)]
        for cuts, chunk in cut_sets:
            for wrapper, cls in ((0, da.DeflateEncoder), (1, da.ZlibEncoder)):
                got = _drive(lambda: cls(io.BytesIO(), da.CompressionOptions(c, l, m), ctx),
                             lambda: ob.Stream(ob.make_opts(c, l, m, wrapper)), data, cuts,
                             flush_at_end=(cuts == [n // 3]), chunk=chunk)
                if wrapper and cuts and cuts[0] == 0:
                    # reference quirk reproduced bit for bit: ZlibEncoder::flush() before the first write
                    # emits the sync block BEFORE the zlib header (writer.rs:274-277 vs :254), which is
                    # not a valid zlib stream; the raw deflate part after the header still is
                    lead = 6 * sum(1 for x in cuts if x == 0)
                    assert got[lead:lead + 2] == b"\x78\x9c"
                    assert inflate_raw(got[:lead] + got[lead + 2:-4]) == data
                elif wrapper:
                    assert zlib.decompress(got) == data
                else:
                    assert inflate_raw(got) == data
    # later data may match across the flush point: the stream with a flush is smaller than two streams
    data = datagen.text_like(20000, 42) * 2  # the second copy lies within the 32 KiB window of the first
    one = _drive(lambda: da.DeflateEncoder(io.BytesIO(), da.Compression.Default, ctx),
                 lambda: ob.Stream(ob.preset(ob.DEFAULT)), data, [20000])
    two = len(ctx.encode(data[:20000])) + len(ctx.encode(data[20000:]))
    assert len(one) < two * 0.7


# writer.rs:570-595 writer_sync: right after flush() the inner writer already holds the stream up to and
# including the sync marker 00 00 FF FF
def test_writer_sync_bytes_are_delivered_at_flush(da, ctx):
    import io
    data = open(os.path.join(FIX, "pg11.txt"), "rb").read()
    split = len(data) // 2
    for cls, wrapper in ((da.DeflateEncoder, 0), (da.ZlibEncoder, 1), (da.GzEncoder, 2)):
        sink = io.BytesIO()
        enc = cls(sink, da.Compression.Default, ctx)
        ref = ob.Stream(ob.preset(ob.DEFAULT, wrapper))
        if wrapper == 2:
            ref.gzip_header(da.BLANK_GZIP_HEADER)  # GzEncoder::new = GzBuilder::new() (writer.rs:340-342)
        enc.write_all(data[:split])
        ref.write_all(data[:split])
        enc.flush()
        ref.flush()
        held = sink.getvalue()
        assert held[-4:] == b"\x00\x00\xff\xff"
        assert held == ref.output()  # the reference's Vec holds the same bytes at this point
        enc.write_all(data[split:])
        ref.write_all(data[split:])
        got = enc.finish().getvalue()
        assert got == ref.finish()
        body = got if wrapper == 0 else (got[2:-4] if wrapper == 1 else got[10:-8])
        assert inflate_raw(body) == data


# tests/test.rs:163-200 issue_47: a sink that takes at most two bytes per write() call
class SmallWriter:
    def __init__(self, small):
        self.buf = bytearray()
        self.small = small
        self.calls = 0

    def write(self, b):
        k = min(len(b), self.small)
        self.buf += b[:k]
        self.calls += 1
        return k


def test_issue_47_short_write_sink(da, ctx):
    w = SmallWriter(2)
    enc = da.ZlibEncoder(w, da.Compression.Fast, ctx)
    enc.flush()  # the reference's test: a flush on a fresh encoder into the small writer must not hang or fail
    ref = ob.Stream(ob.preset(ob.FAST, 1))
    ref.flush()
    assert bytes(w.buf) == ref.output()
    data = datagen.text_like(50000, 7)
    enc.write_all(data)
    ref.write_all(data)
    enc.flush()
    ref.flush()
    assert bytes(w.buf) == ref.output() and w.calls >= len(w.buf) // 2
    enc.finish()
    assert bytes(w.buf) == ref.finish()


# writer.rs:139-152 Drop: an encoder that goes away unfinished finishes its stream into the writer
def test_drop_finishes_the_stream(da, ctx):
    import io
    data = datagen.text_like(30000, 3)
    sink = io.BytesIO()
    enc = da.ZlibEncoder(sink, da.Compression.Default, ctx)
    enc.write_all(data)
    del enc
    assert zlib.decompress(sink.getvalue()) == data
    assert sink.getvalue() == ob.encode(data, level=ob.DEFAULT, wrapper=1)


# Per-message sync flush: hundreds of small incompressible writes, each followed by a flush -- every
# segment costs a block header and a marker on top of its bytes (the output bound has to count them).
def test_many_small_flushed_writes(da, ctx):
    import io
    for count, size, kind in ((50, 100, "rng"), (1000, 30, "rng"), (400, 300, "text")):
        for cls, wrapper in ((da.DeflateEncoder, 0), (da.ZlibEncoder, 1)):
            enc = cls(io.BytesIO(), da.Compression.Default, ctx)
            ref = ob.Stream(ob.preset(ob.DEFAULT, wrapper))
            whole = b""
            for i in range(count):
                piece = datagen.rng_bytes(size, i + 1) if kind == "rng" else datagen.text_like(size, i + 1)
                whole += piece
                enc.write_all(piece)
                ref.write_all(piece)
                enc.flush()
                ref.flush()
            got = enc.finish().getvalue()
            assert got == ref.finish()
            assert (zlib.decompress(got) if wrapper else inflate_raw(got)) == whole


# A long stream with flushes: once the flushed part has passed three windows the handle keeps only the
# 32 KiB window before the last flush point and encodes every new segment against it.
@pytest.mark.parametrize("level", ["default", "fast", "best"])
def test_long_stream_keeps_only_its_window(da, ctx, level):
    import io
    import random
    c, l, m = LV[level]
    rnd = random.Random(11)
    data = datagen.text_like(900_000, 21) + datagen.mixed(300_000, 22) + (datagen.rng_bytes(257, 23) * 800)
    for wrapper, cls in ((0, da.DeflateEncoder), (1, da.ZlibEncoder), (2, da.GzEncoder)):
        enc = cls(io.BytesIO(), da.CompressionOptions(c, l, m), ctx)
        ref = ob.Stream(ob.make_opts(c, l, m, wrapper))
        if wrapper == 2:
            ref.gzip_header(da.BLANK_GZIP_HEADER)
        pos = 0
        mid = None
        while pos < len(data):
            step = rnd.choice([3, 700, 20_000, 33_000, 70_000, 150_000])
            if len(data) - (pos + step) == 1:
                step += 1
            piece = data[pos:pos + step]
            # several write calls per segment, never a 1-byte one
            half = len(piece) // 2
            for part in ((piece[:half], piece[half:]) if half >= 2 and len(piece) - half >= 2 else (piece,)):
                enc.write_all(part)
                ref.write_all(part)
            pos += len(piece)
            if pos < len(data) or rnd.random() < 0.5:
                enc.flush()
                ref.flush()
            if mid is None and pos > 600_000 and wrapper:
                mid = (enc.checksum(), ref.checksum())
        got = enc.finish().getvalue()
        assert got == ref.finish()
        if mid:
            assert mid[0] == mid[1]


# ---- SURVEY section 8 row g: inputs of any length, never-flushed streams in bounded memory (deflate_long.inc) ----
def _set_long(ctx, range_bytes, from_bytes):
    ctx.config(ctx.CFG_RANGE_BYTES, range_bytes)
    ctx.config(ctx.CFG_LONG_FROM, from_bytes)


@pytest.fixture
def small_ranges(da, ctx):
    """16 MiB ranges, every one-shot call of 20 MB or more goes through them (mi355_deflate_ctx_config)"""
    _set_long(ctx, 16 << 20, 20_000_000)
    yield
    _set_long(ctx, 512 << 20, (1 << 30) + 1)


def test_identity_hop_after_first_slide(da, ctx):
    """tests/golden/identity_hop.bin (gen_identity_hop.py): the first block fills exactly at the end of the first window, so the
    hash re-warm (Q1, lz77.rs:628-638) files positions 32768 and 32769 under hashes that are not their bytes'; after the first
    slide the head table's identity entries 0 and 1 (chained_hash_table.rs:197-219) point at exactly those two positions, and
    the third window begins with a copy of their bytes: position 65536 of the reference finds a match of 64 at distance 32768
    through an entry no bucket holds.  The GPU path must produce the oracle's stream (the committed digests) -- at every level
    with a hash, one-shot and as a stream written in two parts."""
    import hashlib
    import json
    import io
    gold = os.path.join(HERE, "golden")
    data = open(os.path.join(gold, "identity_hop.bin"), "rb").read()
    meta = json.load(open(os.path.join(gold, "identity_hop.json")))
    assert hashlib.sha256(data).hexdigest() == meta["input_sha256"]
    for name, (c, l, m) in {"default": (128, 32, 1), "best": (1768, 128, 1), "fast": (1, 0, 0), "greedy128": (128, 0, 0)}.items():
        want = ob.encode(data, opts=ob.make_opts(c, l, m))
        assert hashlib.sha256(want).hexdigest() == meta["streams"][name]["sha256"], name  # the oracle is the committed one
        info = agree(da, ctx, data, c, l, m)
        assert info["q1_rewarm"] == 1, name
        enc = da.DeflateEncoder(io.BytesIO(), da.CompressionOptions(c, l, m), ctx=ctx)
        enc.write(data[:50000])
        enc.write(data[50000:])
        assert enc.finish().getvalue() == want, (name, "written in two parts")
    # and the input with the first three windows moved behind a window of other bytes: no re-warm, no such match
    pre = datagen.text_like(32768, 5)
    agree(da, ctx, pre + data, *LV["default"])


def test_host_call_streamed_in_pieces(da):
    """mi355_deflate_encode of 16 MiB or more works on the input piece by piece as it arrives and hands the finished bytes
    back while the later pieces are worked on (run_streamed; MI355_CFG_HOST_STREAMING): the bytes are those of the single
    pass, i.e. the oracle's -- page-locked buffers (the copy engine carries the pieces) and pageable ones (the runtime's
    copies), raw / zlib / gzip, the levels with a hash, and the inputs the piecewise form hands to the single pass (noise:
    the hash re-warm of the first window; zeros: periodic, the speculative entries do not hold)."""
    import torch
    ctx = da.Context(0)
    try:
        cases = [("text", datagen.text_like(40_000_000, 91), ("default", "fast", "best")),
                 ("mixed", datagen.mixed(30_000_000, 92) + datagen.text_like(9_000_001, 93), ("default",)),
                 ("noise", datagen.rng_bytes(33_000_000, 94), ("default",)),
                 ("zeros", bytes(36_000_000), ("default",))]
        for name, data, levels in cases:
            h_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).pin_memory()
            cap = da.bound(len(data)) + 64
            h_out = torch.empty(cap, dtype=torch.uint8).pin_memory()
            for lv in levels:
                c, l, m = LV[lv]
                for wrapper in (0, 1, 2):
                    if wrapper and lv != "default":
                        continue
                    want = ob.encode(data, opts=ob.make_opts(c, l, m, wrapper)) if wrapper < 2 else ob.encode_gzip(
                        data, da.BLANK_GZIP_HEADER, opts=ob.make_opts(c, l, m, 0))
                    pieces = {}
                    for mode in (1, 0, 2):
                        ctx.config(da.Context.CFG_HOST_STREAMING, mode)
                        h_out.zero_()
                        n = ctx.encode_host_ptr(h_in.data_ptr(), len(data), h_out.data_ptr(), cap, da.CompressionOptions(c, l, m),
                                                wrapper=wrapper)
                        assert bytes(h_out[:n].numpy()) == want, (name, lv, wrapper, mode, n, len(want))
                        pieces[mode] = ctx.info()["host_path"]
                    assert pieces[0] == 0 and pieces[1] == pieces[2] and pieces[1] in (0, da.Context.HOST_PATH_PIECES)
                    if name == "text":
                        assert pieces[1] == da.Context.HOST_PATH_PIECES
                    if name in ("noise", "zeros"):
                        assert pieces[1] == 0
                    ctx.config(da.Context.CFG_HOST_STREAMING, 1)
                    got = ctx.encode(data, da.CompressionOptions(c, l, m), wrapper=wrapper)  # pageable buffers
                    assert got == want, (name, lv, wrapper, "pageable", len(got), len(want))
                    # the pageable call took the streamed form too: in and out through the context's host threads
                    hp = ctx.info()["host_path"]
                    P, I, O = da.Context.HOST_PATH_PIECES, da.Context.HOST_PATH_IN_THREADS, da.Context.HOST_PATH_OUT_THREADS
                    assert hp & I, (name, lv, wrapper, hp)
                    assert (hp & P) == pieces[1], (name, lv, wrapper, hp, pieces)  # (... exactly where the page-locked call did)
                    if (hp & P) or len(want) >= (4 << 20):  # (a few KB of output -- zeros -- take the runtime's copy)
                        assert hp & O, (name, lv, wrapper, hp)
        # an output buffer that is too small for the bound goes the single pass's way and says so only if the bytes do not fit
        data = cases[0][1]
        want = ob.encode(data, opts=ob.make_opts(*LV["default"], 0))
        h_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).pin_memory()
        h_out = torch.empty(len(want) + 8, dtype=torch.uint8).pin_memory()
        n = ctx.encode_host_ptr(h_in.data_ptr(), len(data), h_out.data_ptr(), len(want) + 8, da.Compression.Default)
        assert bytes(h_out[:n].numpy()) == want
        with pytest.raises(da.DeflateError):
            ctx.encode_host_ptr(h_in.data_ptr(), len(data), h_out.data_ptr(), len(want) - 1, da.Compression.Default)
    finally:
        ctx.close()


def test_pageable_host_buffers_through_the_host_threads(da):
    """deflate_bytes(&[u8]) -> Vec<u8> (src/lib.rs:137-147,163) hands over pageable memory: from 4 MiB on the context's host
    threads carry it through page-locked slots (deflate_bounce.inc, MI355_CFG_HOST_BOUNCE) -- same bytes as the runtime's own
    copies, i.e. the oracle's; sizes around the 1 MiB chunks and the 16 MiB from which a call works in pieces; one thread and
    many; page-locked buffers never take the threads; the call's error paths leave the threads idle (a buffer too small)."""
    import numpy as np
    import torch
    P, I, O = da.Context.HOST_PATH_PIECES, da.Context.HOST_PATH_IN_THREADS, da.Context.HOST_PATH_OUT_THREADS
    base = datagen.text_like(24_000_000, 0x9A6E) + datagen.rng_bytes(3_000_000, 0x9A6F) + datagen.mixed(21_000_000, 0x9A70)
    c, l, m = LV["default"]
    for threads in (1, 3, 0):
        ctx = da.Context(0)
        try:
            if threads:
                ctx.config(da.Context.CFG_HOST_THREADS, threads)
            for n in (4 << 20, (4 << 20) - 1, (5 << 20) + 12345, (16 << 20) - 1, 16 << 20, (17 << 20) + 1, len(base)):
                data = base[len(base) - n:]
                want = ob.encode(data, opts=ob.make_opts(c, l, m, 0))
                for bounce in (1, 0):
                    ctx.config(da.Context.CFG_HOST_BOUNCE, bounce)
                    p_in = np.frombuffer(data, dtype=np.uint8).copy()
                    cap = da.bound(n) + 64
                    p_out = np.full(cap, 0xEE, dtype=np.uint8)
                    got_n = ctx.encode_host_ptr(p_in.ctypes.data, n, p_out.ctypes.data, cap, da.Compression.Default)
                    assert bytes(p_out[:got_n]) == want, (threads, n, bounce, got_n, len(want))
                    assert np.all(p_out[got_n:] == 0xEE), (threads, n, bounce)  # nothing written behind the stream
                    hp = ctx.info()["host_path"]
                    if bounce and n >= (4 << 20):
                        assert hp & I, (threads, n, hp)
                        assert bool(hp & O) == (len(want) >= (4 << 20) or bool(hp & P)), (threads, n, hp, len(want))
                        assert not (hp & P) or n >= (16 << 20), (threads, n, hp)  # (pieces from 16 MiB on -- where the data lets them)
                    else:
                        assert not hp & (I | O), (threads, n, bounce, hp)
            # (a context of its own for what follows: after data whose speculative parse failed -- the mixed tail above -- a context
            # leaves the piecewise form alone for its next sixteen calls)
            ctx.close()
            ctx = da.Context(0)
            if threads:
                ctx.config(da.Context.CFG_HOST_THREADS, threads)
            # zlib and gzip frames through the threads (the header bytes of a stream that left in pieces are put in on the host);
            # text and noise: data the piecewise form takes (the mixed tail has runs its speculative parse gives up on)
            data = base[:27_000_000]
            for wrapper in (1, 2):
                want = ob.encode(data, opts=ob.make_opts(c, l, m, 1)) if wrapper == 1 else ob.encode_gzip(
                    data, da.BLANK_GZIP_HEADER, opts=ob.make_opts(c, l, m, 0))
                assert ctx.encode(data, da.Compression.Default, wrapper=wrapper) == want, wrapper
                assert ctx.info()["host_path"] == P | I | O
            # page-locked buffers: the copy engines directly
            h_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).pin_memory()
            h_out = torch.empty(da.bound(len(data)) + 64, dtype=torch.uint8).pin_memory()
            got_n = ctx.encode_host_ptr(h_in.data_ptr(), len(data), h_out.data_ptr(), h_out.numel(), da.Compression.Default)
            assert bytes(h_out[:got_n].numpy()) == ob.encode(data, opts=ob.make_opts(c, l, m, 0))
            assert ctx.info()["host_path"] == P
            # pageable in, page-locked out and the other way round
            p_in = np.frombuffer(data, dtype=np.uint8).copy()
            got_n = ctx.encode_host_ptr(p_in.ctypes.data, len(data), h_out.data_ptr(), h_out.numel(), da.Compression.Default)
            assert bytes(h_out[:got_n].numpy()) == ob.encode(data, opts=ob.make_opts(c, l, m, 0)) and ctx.info()["host_path"] == P | I
            p_out = np.zeros(h_out.numel(), dtype=np.uint8)
            got_n = ctx.encode_host_ptr(h_in.data_ptr(), len(data), p_out.ctypes.data, p_out.size, da.Compression.Default)
            assert bytes(p_out[:got_n]) == ob.encode(data, opts=ob.make_opts(c, l, m, 0)) and ctx.info()["host_path"] == P | O
            # an output buffer that cannot hold the stream: an error, and the next call works
            small = np.zeros(1_000_000, dtype=np.uint8)
            with pytest.raises(da.DeflateError):
                ctx.encode_host_ptr(p_in.ctypes.data, len(data), small.ctypes.data, small.size, da.Compression.Default)
            assert ctx.encode(data[:9_000_000], da.Compression.Default) == ob.encode(data[:9_000_000], opts=ob.make_opts(c, l, m, 0))
            # the levels without a hash read the whole input at once (rle, huffman_only), noise re-warms (Q1), zeros fall back
            for lv, d2 in (("rle", bytes(20_000_000) + base[:5_000_000]), ("huffman_only", base[:18_000_000]),
                           ("default", datagen.rng_bytes(19_000_000, 0x9A71)), ("default", bytes(33_000_000)), ("best", base[:20_000_000])):
                cc, ll, mm = LV[lv]
                assert ctx.encode(d2, da.CompressionOptions(cc, ll, mm)) == ob.encode(d2, opts=ob.make_opts(cc, ll, mm, 0)), lv
        finally:
            ctx.close()


def test_multi_gpu_pinned_host_range_at_a_level_without_a_hash(da):
    """Round-5 advisor finding: a rank's host range of 16 MiB or more arrives in pieces on the copy stream, and only the hashing
    levels waited for it -- rle() and huffman_only() read the staging buffer with no ordering against the copy.  Pageable
    buffers hid it (their copies are synchronous); page-locked ones must give the oracle's bytes too, call after call with
    different data in the same staging buffer."""
    import torch
    m = da.MultiGpu([0, 0])
    try:
        for k, lv in enumerate(("rle", "huffman_only", "rle", "default")):
            data = (datagen.text_like(20_000_000, 0xAD0 + k) + bytes(17_000_000) + datagen.mixed(13_000_000, 0xAE0 + k))[k:]
            c, l, mt = LV[lv]
            want = ob.encode(data, opts=ob.make_opts(c, l, mt, 0))
            h_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).pin_memory()
            cap = da.bound(len(data)) + 64
            h_out = torch.zeros(cap, dtype=torch.uint8).pin_memory()
            n = m.encode_host_ptr(h_in.data_ptr(), len(data), h_out.data_ptr(), cap, da.CompressionOptions(c, l, mt))
            assert bytes(h_out[:n].numpy()) == want, (lv, n, len(want))
            assert m.encode(data, da.CompressionOptions(c, l, mt)) == want, (lv, "pageable")
    finally:
        m.close()


def test_rows_of_records_take_the_permuted_pair_table(da, ctx):
    """Rows of fixed-length records put the lanes of a wave -- neighbours in a hash bucket, one row apart -- on a few of the LDS's
    banks; k_sort marks such epochs and k_match3_swz walks them with the pair table's 8-byte words permuted (PairWinT<true>).
    Layout only: the bytes are the oracle's at every level, for row lengths that hit one bank (256, 1024), four (96), eight (48)
    and sixteen (40: not marked), for epochs walked whole (48 MB) and in parts (3 MB), and where marked and unmarked epochs
    alternate (records between text)."""
    import numpy as np

    def records(n, width, seed):
        r = np.random.default_rng(seed)
        rows = n // width + 1
        a = np.tile(r.integers(0, 256, size=width, dtype=np.uint8), (rows, 1))
        a[:, 4:8] = np.arange(rows, dtype=np.uint32).view(np.uint8).reshape(rows, 4)
        cols = r.choice(np.arange(8, width), size=max(1, width // 8), replace=False)
        a[:, cols] = r.integers(0, 16, size=(rows, len(cols)), dtype=np.uint8)
        return a.reshape(-1)[:n].tobytes()
    for width, n in ((96, 3_000_000), (256, 3_000_001), (1024, 2_500_000), (48, 2_000_000), (40, 2_000_000), (97, 1_500_000), (64, 2_000_000), (128, 2_200_000)):
        data = records(n, width, 0x5EC0 + width)
        for lv in ("default", "best", "fast"):
            agree(da, ctx, data, *LV[lv])
    mix = b"".join([datagen.text_like(700_000, 0x5EC1), records(1_300_000, 96, 0x5EC2), datagen.text_like(300_001, 0x5EC3),
                    records(900_000, 256, 0x5EC4), datagen.rng_bytes(200_000, 0x5EC5), records(600_000, 512, 0x5EC6)])
    for lv in ("default", "best"):
        agree(da, ctx, mix, *LV[lv])
    big = records(48_000_000, 96, 0x5EC7)[:-5] + datagen.text_like(4_000_000, 0x5EC8) + records(9_000_000, 256, 0x5EC9)
    for lv in ("default", "best"):
        c, l, m = LV[lv]
        assert ctx.encode(big, da.CompressionOptions(c, l, m)) == ob.encode(big, opts=ob.make_opts(c, l, m, 0)), lv


def test_long_input_walked_in_ranges(da, ctx, small_ranges):
    """One call, several ranges (the phases of the sharded encode one after the other on one GPU, the bit position and
    the unfinished block carried across): the bytes of a single-range call, i.e. of the oracle -- text, zeros (a range
    is a few blocks), noise (stored blocks across the seams, Q1), a mix; every level; raw, zlib and gzip; host and
    device buffers."""
    import torch
    cases = [("text", datagen.text_like(52_000_000, 77), ("default", "fast", "best")),
             ("zeros", bytes(40_000_000), ("default", "rle")),
             ("noise", datagen.rng_bytes(36_000_000, 78), ("default",)),
             ("mixed", datagen.mixed(34_000_000, 79) + datagen.text_like(8_000_000, 80), ("default", "huffman_only"))]
    for name, data, levels in cases:
        for lv in levels:
            c, l, m = LV[lv]
            for wrapper in (0, 1, 2):
                if wrapper and lv not in ("default",):
                    continue
                want = ob.encode(data, opts=ob.make_opts(c, l, m, wrapper)) if wrapper < 2 else ob.encode_gzip(
                    data, da.BLANK_GZIP_HEADER, opts=ob.make_opts(c, l, m, 0))
                got = ctx.encode(data, da.CompressionOptions(c, l, m), wrapper=wrapper)
                assert got == want, (name, lv, wrapper, len(got), len(want))
                assert ctx.info()["passes"] >= 2, "the call was meant to take several ranges"
            # device buffers
            d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
            cap = da.bound(len(data)) + 8
            d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
            n = ctx.encode_device(d_in.data_ptr(), len(data), d_out.data_ptr(), cap, da.CompressionOptions(c, l, m))
            assert bytes(d_out[:n].cpu().numpy()) == ob.encode(data, opts=ob.make_opts(c, l, m, 0)), (name, lv, "device")
            if lv == "default":
                # a sync-flush chunk goes through ranges as well (what a fresh encoder has written after write_all + flush():
                # header, every block non-final, the marker 00 00 FF FF, no trailer), host and device buffers
                for wrapper in (0, 1):
                    ref = ob.Stream(ob.make_opts(c, l, m, wrapper))
                    ref.write_all(data)
                    ref.flush()
                    want = ref.output()
                    got = ctx.encode(data, da.CompressionOptions(c, l, m), wrapper=wrapper, flush=da.FLUSH_SYNC)
                    assert got == want and got.endswith(b"\x00\x00\xff\xff"), (name, "sync chunk", wrapper, len(got), len(want))
                    assert ctx.info()["passes"] >= 2
                d_out.zero_()
                n = ctx.encode_device(d_in.data_ptr(), len(data), d_out.data_ptr(), cap, da.CompressionOptions(c, l, m),
                                      flush=da.FLUSH_SYNC)
                ref = ob.Stream(ob.make_opts(c, l, m, 0))
                ref.write_all(data)
                ref.flush()
                assert bytes(d_out[:n].cpu().numpy()) == ref.output(), (name, "sync chunk on the device")


def test_never_flushed_stream_is_bounded(da, ctx, small_ranges):
    """write() hands a range over whenever one has filled up with its margin behind it (output independent of the write
    chunking, lib.rs:408-433): the handle never holds more than a range, the margin, a window and a write; the bytes
    are the oracle's; a flush() later ends the ranges with the sync marker and the stream goes on as a flushed one."""
    import io
    import random
    data = datagen.text_like(70_000_000, 91)
    L = da.load()
    for wrapper, cls, lv in ((0, da.DeflateEncoder, "default"), (1, da.ZlibEncoder, "default"), (2, da.GzEncoder, "fast")):
        c, l, m = LV[lv]
        for flush_at in (None, 45_000_000):
            rnd = random.Random(wrapper)
            enc = cls(io.BytesIO(), da.CompressionOptions(c, l, m), ctx)
            ref = ob.Stream(ob.make_opts(c, l, m, wrapper))
            if wrapper == 2:
                ref.gzip_header(da.BLANK_GZIP_HEADER)
            pos, held, mid = 0, 0, None
            while pos < len(data):
                step = rnd.choice([1, 5000, 65_536, 1_000_003, 3_500_000])
                if flush_at and pos < flush_at <= pos + step:
                    step = flush_at - pos
                enc.write_all(data[pos:pos + step])
                ref.write_all(data[pos:pos + step])
                pos += step
                held = max(held, L.mi355_deflate_stream_held_bytes(enc._s))
                if flush_at and pos == flush_at:
                    enc.flush()
                    ref.flush()
                    assert enc._w.getvalue().endswith(b"\x00\x00\xff\xff")
                if mid is None and pos > 40_000_000 and wrapper:
                    mid = (enc.checksum(), ref.checksum())
            got = enc.finish().getvalue()
            assert got == ref.finish(), (wrapper, flush_at)
            if mid:
                assert mid[0] == mid[1]
            # range 16 MiB + margin 16 MiB + look-ahead + window + the largest write, not the 70 MB of the stream
            assert held < 42_000_000, held


def test_flushed_stream_is_bounded_between_flushes(da, ctx, small_ranges):
    """What is written after a flush() is handed over in ranges as well: the first range begins AT the flush point (anywhere in
    a window, the parse entering exactly there, the blocks counting from there) with the hash quirks of the write calls around
    the flush -- a 1-byte write after it leaves a position out of the chains and files two a byte late (lz77.rs:601-614) -- and
    the handle holds a range and its margin, not what has gathered since the flush (the reference: O(window) whatever the
    flush pattern, compress.rs:96-124).  Bytes of the oracle driven with the same calls."""
    import io
    import random
    data = datagen.text_like(50_000_000, 0x71) + datagen.mixed(20_000_000, 0x72) + datagen.text_like(40_000_000, 0x73)
    L = da.load()
    flushes = [7_000_123, 7_000_125, 57_345_679, 57_400_000]  # (two of them two bytes / a few KB apart)
    for wrapper, cls, lv, first_after in ((0, da.DeflateEncoder, "default", 1), (1, da.ZlibEncoder, "best", 70_001),
                                          (2, da.GzEncoder, "fast", 2)):
        c, l, m = LV[lv]
        rnd = random.Random(wrapper + first_after)
        enc = cls(io.BytesIO(), da.CompressionOptions(c, l, m), ctx)
        ref = ob.Stream(ob.make_opts(c, l, m, wrapper))
        if wrapper == 2:
            ref.gzip_header(da.BLANK_GZIP_HEADER)
        pos, held, after = 0, 0, False
        todo = list(flushes)
        while pos < len(data):
            step = first_after if after else rnd.choice([5000, 65_536, 1_000_003, 3_500_000])
            after = False
            if todo and pos < todo[0] <= pos + step:
                step = todo[0] - pos
            step = min(step, len(data) - pos)
            enc.write_all(data[pos:pos + step])
            ref.write_all(data[pos:pos + step])
            pos += step
            held = max(held, L.mi355_deflate_stream_held_bytes(enc._s))
            if todo and pos == todo[0]:
                todo.pop(0)
                enc.flush()
                ref.flush()
                assert enc._w.getvalue().endswith(b"\x00\x00\xff\xff")
                after = True
        got = enc.finish().getvalue()
        want = ref.finish()
        assert got == want, (wrapper, lv, first_after, len(got), len(want))
        # range 16 MiB + margin 16 MiB + look-ahead + window + the largest write -- not the 50 MB / 52 MB between the flushes
        assert held < 42_000_000, held


def test_flushed_stream_behind_an_early_flush(da, ctx, small_ranges):
    """The same behind a flush point inside the first three windows, where a stream's start has rules of its own: the range that
    begins there holds the stream from its start and takes over what the flush calls of the stream passed to the single pass --
    the flush points that re-warm the hash, the Q1 re-warm a flush call found (noise at the start: block 0 fills inside the
    first window) -- or finds Q1 itself when the block that fills inside the first window begins at the flush point."""
    import io
    L = da.load()
    text = datagen.text_like(41_000_000, 0x83)
    # Input on which Q1 SHOWS: noise whose first 31 744 tokens end inside the first window (the re-warm files the two positions
    # behind that block under hashes of the stream's first two bytes), then a short copy of the noise at every position from
    # 31 700 to 32 100, each behind a separator: the copy of a re-warmed position finds no candidate there (the oracle: a match
    # of 7 at distance 8 into the copy before it instead of 8 at the distance of the noise -- checked with tools/tokdump.py).
    noise = datagen.rng_bytes(33_000, 0x91)
    seps = datagen.rng_bytes(400, 0x92)
    copies = b"".join(noise[k:k + 8] + seps[k - 31_700:k - 31_699] for k in range(31_700, 32_100))
    q1_early = noise[:32_200] + copies + text      # flush at 5: the block that fills inside the first window begins there
    q1_given = noise + copies + text               # flush at 33 000: a flush call has found Q1, the copies come behind the flush
    cases = [(q1_early, [5], 1, "default"),
             (q1_early, [5], 2, "best"),
             (q1_given, [33_000], 1, "default"),
             (q1_given, [33_000], 70_000, "fast"),
             (q1_given, [40_000, 40_002], 1, "fast"),
             (text, [20_000], 1, "default"),           # a flush point that re-warms the hash, a 1-byte write behind it
             (text, [2, 3], 70_001, "best"),
             (text, [98_303], 1, "default")]
    for data, flushes, first_after, lv in cases:
        c, l, m = LV[lv]
        enc = da.ZlibEncoder(io.BytesIO(), da.CompressionOptions(c, l, m), ctx)
        ref = ob.Stream(ob.make_opts(c, l, m, 1))
        pos, held, after, todo = 0, 0, False, list(flushes)
        while pos < len(data):
            step = first_after if after else 3_000_000
            after = False
            if todo and pos < todo[0] <= pos + step:
                step = todo[0] - pos
            step = min(step, len(data) - pos)
            enc.write_all(data[pos:pos + step])
            ref.write_all(data[pos:pos + step])
            pos += step
            held = max(held, L.mi355_deflate_stream_held_bytes(enc._s))
            if todo and pos == todo[0]:
                todo.pop(0)
                enc.flush()
                ref.flush()
                after = True
        got = enc.finish().getvalue()
        want = ref.finish()
        assert got == want, (flushes, first_after, lv, len(got), len(want))
        assert held < 40_000_000, held  # (not the 41-44 MB behind the flush)


@pytest.mark.parametrize("flush_after", [None, (256 << 20) + 12345])
def test_stream_beyond_4gib_without_flush(da, ctx, flush_after):
    """A ZlibEncoder fed 4.25 GiB (64 MiB of web text, 68 times over) and never flushed: positions beyond 2^32, 512 MiB
    ranges handed over as they fill, the handle holds a range and its margin, not the stream; the stream inflates to
    the input (checked piece by piece) and ends in its Adler-32.  The same with ONE flush() a quarter GiB in and a 1-byte
    write behind it: 4 GiB follow the flush point -- more than a single pass takes, refused before round 4 -- in ranges
    that begin AT the flush point (anywhere in a window), the handle as small as without the flush."""
    import io
    block = datagen.webtext(64 << 20)
    reps = 68
    _set_long(ctx, 512 << 20, (1 << 30) + 1)
    L = da.load()

    class Check:  # the inner writer: inflates what it is given and compares it with the input, keeps nothing
        def __init__(self):
            self.d = zlib.decompressobj()
            self.pos = 0
            self.n = 0

        def write(self, b):
            self.n += len(b)
            out = self.d.decompress(b)
            o = 0
            while o < len(out):
                k = min(len(out) - o, len(block) - self.pos % len(block))
                assert out[o:o + k] == block[self.pos % len(block):self.pos % len(block) + k], self.pos
                o += k
                self.pos += k

    sink = Check()
    enc = da.ZlibEncoder(sink, da.Compression.Default, ctx)
    held = 0
    written = 0
    for _ in range(reps):
        o = 0
        while o < len(block):
            k = 32 << 20
            if flush_after is not None and written < flush_after < written + k:
                k = flush_after - written
            elif flush_after is not None and written == flush_after:
                enc.flush()
                k = 1
            k = min(k, len(block) - o)
            enc.write_all(block[o:o + k])
            o += k
            written += k
            held = max(held, L.mi355_deflate_stream_held_bytes(enc._s))
    enc.finish()
    assert sink.d.eof and sink.pos == reps * len(block)  # (zlib has checked the Adler-32 of all 4.25 GiB)
    assert held < (512 << 20) + (96 << 20), held
    assert sink.n < 0.5 * sink.pos


def test_input_beyond_2_31(da, ctx):
    """2^31 + 2^20 bytes of the web-text input in one call: positions beyond 2^31, 512 MiB ranges, against the oracle's
    digest (tests/golden/long_digest.json, gen_long_digest.py)."""
    import hashlib
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "long_digest.json")))["digest"]
    _set_long(ctx, 512 << 20, (1 << 30) + 1)
    data = datagen.webtext(gold["in_len"])
    assert hashlib.sha256(data).hexdigest() == gold["in_sha256"]
    got = ctx.encode(data, da.Compression.Default)
    assert len(got) == gold["out_len"] and hashlib.sha256(got).hexdigest() == gold["out_sha256"]
    assert ctx.info()["passes"] >= 4


# tests/test.rs:113-123 issue_26 (write, flush, one-byte write, write, drop) and its relatives: a one-byte
# write right after a flush files one position less and two a byte late (lz77.rs:605-614), a flush after
# one or two bytes leaves the first positions out of the chains and re-warms the hash (lz77.rs:606,628-638).
# Periodic data makes every misfiled position show up as a different distance.
def test_issue_26_and_the_write_patterns_around_a_flush(da, ctx):
    import io
    per = datagen.rng_bytes(300, 3) * 500
    txt = datagen.text_like(400_000, 31)
    scripts = [
        [b"\0", "F", b"\0", b"\0\0"],                                     # issue_26 itself
        [b"x" * 1000, "F", b"y", b"z" * 100],
        [per[:5000], "F", per[5000:5001], per[5001:90000]],
        [per[:5000], "F", per[5000:5001], "F", per[5001:5002], per[5002:90000]],
        [per[:1], "F", per[1:90000]],
        [per[:2], "F", per[2:90000]],
        [per[:1], "F", per[1:2], "F", per[2:90000]],
        [per[:2], "F", per[2:3], per[3:90000]],
        [per[:40000], "F", per[40000:40001], per[40001:40900], "F", per[40900:120000]],
        [txt[:250_000], "F", txt[250_000:250_001], per[:60000], "F", txt[250_001:250_002], txt[250_002:]],  # past the first windows
        [txt[:150_000], "F", txt[150_000:200_000], "F", per[:1], per[1:50000], "F", per[50000:50001], "F", per[50001:120000]],
        [per[:30000], "F", per[30000:30001], "F", per[30001:30002], "F", per[30002:30003], per[30003:30004], "F", per[30004:100000]],
        [per[:30000], "F", per[30000:30001], per[30001:30002], "F", per[30002:30003], "F", per[30003:100000]],
        # three [flush, write(1), write(10)] cycles past 96 KiB, then a write that makes the next flush trim the history:
        # the trim keeps one flush point but every later skew point and hole (round 2 refused this at finish())
        [txt[:100_000], "F", txt[100_000:100_001], txt[100_001:100_011], "F", txt[100_011:100_012], txt[100_012:100_022], "F",
         txt[100_022:100_023], txt[100_023:100_033], "F", txt[100_033:132_033], "F", txt[132_033:200_000]],
        [per[:100_000], "F"] + sum(([per[100_000 + 11 * k:100_001 + 11 * k], per[100_001 + 11 * k:100_011 + 11 * k], "F"]
                                    for k in range(6)), []) + [per[100_066:140_000], "F", per[140_000:200_000]],
    ]
    refused = 0
    for lv in ("default", "fast", "best"):
        c, l, m = LV[lv]
        for script in scripts:
            for cls, wrapper in ((da.DeflateEncoder, 0), (da.ZlibEncoder, 1)):
                enc = cls(io.BytesIO(), da.CompressionOptions(c, l, m), ctx)
                ref = ob.Stream(ob.make_opts(c, l, m, wrapper))
                whole = b""
                try:
                    for op in script:
                        if op == "F":
                            enc.flush()
                            ref.flush()
                        else:
                            enc.write_all(op)
                            ref.write_all(op)
                            whole += op
                except da.DeflateError as e:
                    # the one pattern that is refused instead of reproduced: two flushes one or two bytes apart
                    assert e.code == da.E_UNSUPPORTED
                    enc._done = True
                    refused += 1
                    continue
                got = enc.finish().getvalue()
                assert got == ref.finish(), (lv, [x if x == "F" else len(x) for x in script], wrapper)
                assert (zlib.decompress(got) if wrapper else inflate_raw(got)) == whole
    assert refused == 0  # (round 1 refused two flushes one or two bytes apart)


def test_tiny_flush_gaps(da, ctx):
    """Writes of 1-4 bytes between sync flushes -- at the start of a stream, inside and beyond the first window,
    around the window edge -- against the oracle: the reference re-adds, skips and re-warms hash entries in
    these calls (lz77.rs:601-638), and nothing of it may be refused or come out differently
    (a slice of tools/fuzz_flush_gaps.py; round 1 refused two flushes one or two bytes apart)."""
    import io
    import random
    n = 140000
    for seed in list(range(1, 90)) + [289, 342, 2133]:
        rnd = random.Random(seed)
        kind = rnd.choice(["per", "text", "zeros", "rng"])
        data = {"per": (datagen.rng_bytes(rnd.choice([1, 3, 300, 4099]), seed) * n)[:n], "text": datagen.text_like(n, seed),
                "zeros": bytes(n), "rng": datagen.rng_bytes(n, seed)}[kind]
        pre = rnd.choice([0, 1, 2, 3, 4, 100, 5000, 30000, 32765, 32766, 32767, 32768, 32769, 32770, 40000, 65535, 65536,
                          65537, 70000])
        ops, pos = ([pre] if pre else []), pre
        for _ in range(rnd.randrange(1, 9)):
            if rnd.random() < 0.45:
                ops.append("F")
            else:
                k = rnd.choice([1, 1, 1, 2, 2, 3, 4, rnd.randrange(1, 2000)])
                ops.append(k)
                pos += k
        tail = rnd.choice([0, 1, 2, 50, 3000, 70000])
        if tail:
            ops.append(min(tail, n - pos))
        lv = rnd.choice(["fast", "default", "best", "rle"])
        c, l, m = LV[lv]
        wrapper = rnd.choice([0, 1])
        enc = (da.ZlibEncoder if wrapper else da.DeflateEncoder)(io.BytesIO(), da.CompressionOptions(c, l, m), ctx)
        ref = ob.Stream(ob.make_opts(c, l, m, wrapper))
        p = 0
        for op in ops:
            if op == "F":
                enc.flush()
                ref.flush()
            else:
                enc.write_all(data[p:p + op])
                ref.write_all(data[p:p + op])
                p += op
        assert enc.finish().getvalue() == ref.finish(), (seed, kind, lv, wrapper, ops)
    # the pattern that showed the identity entries of later epochs: a one-byte write after a flush, period-3 data
    per3 = (datagen.rng_bytes(3, 1) * n)[:n]
    for F in (40000, 65535, 65536, 70000, 100000):
        enc = da.DeflateEncoder(io.BytesIO(), da.CompressionOptions(*LV["fast"]), ctx)
        ref = ob.Stream(ob.make_opts(*LV["fast"], 0))
        for e in (enc, ref):
            e.write_all(per3[:F])
            e.flush()
            e.write_all(per3[F:F + 1])
            e.write_all(per3[F + 1:])
        assert enc.finish().getvalue() == ref.finish(), F


# ---- gzip wrapper, CRC-32 on the GPU (feature "gzip": lib.rs:242-286, writer.rs:293-467) ----
def test_crc32_device(da, ctx):
    import torch
    for n in (0, 1, 3, 4, 5, 15, 16, 17, 511, 512, 513, 131071, 131072, 131073, 1_000_003, 50_000_000):
        data = datagen.rng_bytes(n, n % 1000 + 1) if n < 2_000_000 else datagen.text_like(n, 5)
        t = torch.frombuffer(bytearray(data) or bytearray(1), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        assert ctx.crc32_device(t.data_ptr(), n) == zlib.crc32(data), n
    # an unaligned device pointer takes the byte-wise staging path
    data = datagen.rng_bytes(100003, 9)
    t = torch.frombuffer(bytearray(b"\0" + data), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    assert ctx.crc32_device(t.data_ptr() + 1, len(data)) == zlib.crc32(data)


def test_gzip_one_shot(da, ctx):
    import gzip
    comment = da.gzip_header(comment=b"Comment")
    for data in (b"", b"a", open(os.path.join(FIX, "pg11.txt"), "rb").read(), datagen.rng_bytes(70000, 3),
                 datagen.text_like(3_000_000, 8), bytes(70000)):
        for hdr in (None, comment, da.gzip_header(filename=b"x.txt", extra=b"ab", mtime=12345)):
            z = da.deflate_bytes_gzip_conf(data, da.Compression.Default, hdr, ctx)
            assert z == ob.encode_gzip(data, hdr or da.BLANK_GZIP_HEADER, level=ob.DEFAULT)
            assert gzip.decompress(z) == data
    assert da.deflate_bytes_gzip(b"This is some test data", ctx) == ob.encode_gzip(b"This is some test data",
                                                                                   da.BLANK_GZIP_HEADER)


# writer.rs:473-491 gzip_writer + flush inside a gzip stream (GzEncoder::flush = inner.flush, :446-455)
def test_gzip_writer(da, ctx):
    import gzip
    import io
    data = open(os.path.join(FIX, "pg11.txt"), "rb").read()
    hdr = da.gzip_header(comment=b"Comment")
    enc = da.GzEncoder.from_builder(hdr, io.BytesIO(), da.CompressionOptions.default(), ctx)
    ref = ob.Stream(ob.preset(ob.DEFAULT, 2))
    ref.gzip_header(hdr)
    for part in (data[:len(data) // 2], data[len(data) // 2:]):
        enc.write_all(part)
        ref.write_all(part)
    assert enc.checksum() == zlib.crc32(data) == ref.checksum()
    z = enc.finish().getvalue()
    assert z == ref.finish()
    assert gzip.decompress(z) == data
    for cuts in ([40000], [0, 50000]):
        enc = da.GzEncoder(io.BytesIO(), da.Compression.Default, ctx)
        ref = ob.Stream(ob.preset(ob.DEFAULT, 2))
        ref.gzip_header(da.BLANK_GZIP_HEADER)
        prev = 0
        for c in cuts:
            enc.write_all(data[prev:c])
            ref.write_all(data[prev:c])
            enc.flush()
            ref.flush()
            prev = c
        enc.write_all(data[prev:])
        ref.write_all(data[prev:])
        assert enc.finish().getvalue() == ref.finish()


# writer.rs:537-571 writer_reset, writer_reset_zlib (+ gzip): SURVEY section 8 f2 reset()
def test_reset(da, ctx):
    import io
    data = open(os.path.join(FIX, "pg11.txt"), "rb").read()
    for cls, wrapper in ((da.DeflateEncoder, 0), (da.ZlibEncoder, 1), (da.GzEncoder, 2)):
        enc = cls(io.BytesIO(), da.CompressionOptions.default(), ctx)
        ref = ob.Stream(ob.preset(ob.DEFAULT, wrapper))
        if wrapper == 2:
            ref.gzip_header(da.BLANK_GZIP_HEADER)
        enc.write_all(data)
        ref.write_all(data)
        res1 = enc.reset(io.BytesIO()).getvalue()
        assert res1 == ref.reset()
        if wrapper == 2:
            ref.gzip_header(da.BLANK_GZIP_HEADER)
        enc.write_all(data)
        ref.write_all(data)
        res2 = enc.finish().getvalue()
        assert res2 == ref.finish()
        assert res1 == res2  # the reference's own assertion
    enc = da.DeflateEncoder(io.BytesIO(), da.Compression.Default, ctx)
    enc.write_all(data[:50000])
    enc.flush()
    a = enc.reset(io.BytesIO()).getvalue()
    enc.write_all(data[50000:])
    b = enc.finish().getvalue()
    assert zlib.decompressobj(-15).decompress(a) == data[:50000]
    assert b == ob.encode(data[50000:], level=ob.DEFAULT)


# randomized differential test (tools/fuzz_gpu.py: data kind, size, options, wrapper, write / flush / reset
# script all drawn from the seed); 1500 seeds were run on the MI355X by hand, a slice of them runs here
def test_randomized_streams(da, ctx):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_gpu
    tally = {}
    for seed in range(1, 161):
        r = fuzz_gpu.one(seed, ctx)
        assert not r.startswith("DIFF"), "seed %d: %s" % (seed, r)
        tally[r] = tally.get(r, 0) + 1
    assert tally.get("ok", 0) >= 140


# tests/test.rs:78-91 issue_44 (26 214 400 bytes, 99.99 % zeros with sparse disturbances: one giant hash bucket per
# epoch -- the worst case of the sorted walk and of k_sort's hot digit), inflated, against the oracle
def test_issue_44_on_the_gpu(da, ctx):
    data = zlib.decompress(open(os.path.join(FIX, "issue_44.zlib"), "rb").read())
    assert len(data) == 26214400
    for lv in ("default", "best", "fast"):
        c, l, m = LV[lv]
        want = ob.encode(data, opts=ob.make_opts(c, l, m, 1))
        got = ctx.encode(data, da.CompressionOptions(c, l, m), wrapper=1)
        assert got == want, lv
        assert zlib.decompress(got) == data


# k_sort's two ways to rank a key among the wave's keys of its digit -- ballots (MI355_CFG_SORT_RANKS 0) and returning LDS
# atomics (1, the default on a device that passes k_lds_order_test) -- must give the same sorted arrays, hence the same
# stream: text, runs of one byte, noise, a partial last epoch, a hash re-warm (Q1).
def test_sort_modes_agree(da):
    inputs = [datagen.text_like(3_000_000, 0x5157), bytes(700_001), datagen.mixed(2_500_000, 0x77), datagen.rng_bytes(300_003, 5),
              (b"ab" * 40000 + datagen.text_like(200_000, 9))[:250_017], open(os.path.join(FIX, "pg11.txt"), "rb").read()]
    for mode in (0, 1, None):
        c = da.Context(0)
        try:
            if mode is not None:
                c.config(c.CFG_SORT_RANKS, mode)
            for data in inputs:
                for lv in ("default", "fast", "best"):
                    agree(da, c, data, *LV[lv])
        finally:
            c.close()


# The ranks from LDS atomics rest on the order in which the hardware serves the lanes of one atomic -- sampled by a
# self-test when a context is made, and CHECKED on the data of every encode: k_match3 looks at every entry of a hash
# bucket and the one before it.  The test build of the library (make debug, -DMI355_DEBUG_HOOKS) can make k_sort swap
# two neighbours of a bucket, as a device with another order would: the encode must notice, sort again with ballot
# ranks, give the oracle's bytes, and the context must stay on ballots.
def test_sort_order_is_checked_on_the_data():
    import subprocess
    pkg = os.path.join(ROOT, "deflate-rs_amd")
    lib = os.path.join(pkg, "libmi355deflate_dbg.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", pkg, "-s", "debug"])
    L = C.CDLL(lib)
    import deflate_amd as da
    u8p = C.POINTER(C.c_uint8)
    L.mi355_deflate_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.mi355_deflate_ctx_destroy.argtypes = [C.c_void_p]
    L.mi355_deflate_ctx_destroy.restype = None
    L.mi355_deflate_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(da.Opts), u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mi355_deflate_bound.argtypes = [C.c_size_t]
    L.mi355_deflate_bound.restype = C.c_size_t
    L.mi355_debug_break_sort.argtypes = [C.c_void_p, C.c_int]
    L.mi355_debug_sort_ranks.argtypes = [C.c_void_p]
    L.mi355_shard_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint64, C.c_uint64,
                                    C.POINTER(da.Opts), C.c_void_p, C.POINTER(C.c_void_p)]
    L.mi355_shard_end.argtypes = [C.c_void_p]
    L.mi355_shard_end.restype = None

    def encode(h, data, opts):
        cap = L.mi355_deflate_bound(len(data))
        out = (C.c_uint8 * cap)()
        n = C.c_size_t(0)
        rc = L.mi355_deflate_encode(h, data, len(data), C.byref(opts), out, cap, C.byref(n))
        assert rc == 0, rc
        return bytes(out[:n.value])

    for data, lv in ((datagen.text_like(1_500_000, 0xB0), "default"), (datagen.text_like(40_000_000, 0xB1), "default"),
                     (datagen.mixed(900_000, 0xB2), "best")):
        c, l, m = LV[lv]
        want = ob.encode(data, opts=ob.make_opts(c, l, m, 0))
        opts = da.CompressionOptions(c, l, m).to_c()
        h = C.c_void_p()
        assert L.mi355_deflate_ctx_create(0, C.byref(h)) == 0
        try:
            if L.mi355_debug_sort_ranks(h) != 1:
                pytest.skip("this device sorts with ballot ranks: nothing to break")
            assert encode(h, data, opts) == want  # the test build, unbroken
            assert L.mi355_debug_sort_ranks(h) == 1
            assert L.mi355_debug_break_sort(h, 1) == 0
            assert encode(h, data, opts) == want, "the order check did not catch the broken sort"
            assert L.mi355_debug_sort_ranks(h) == 0, "the context did not fall back to ballot ranks"
            assert encode(h, data, opts) == want
        finally:
            L.mi355_deflate_ctx_destroy(h)
    # the phases of a sharded encode check as well (mi355_shard_begin)
    import torch
    data = datagen.text_like(3_000_000, 0xB3)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    h = C.c_void_p()
    assert L.mi355_deflate_ctx_create(0, C.byref(h)) == 0
    try:
        assert L.mi355_debug_break_sort(h, 1) == 0
        sh = C.c_void_p()
        o = da.CompressionOptions.default().to_c()
        assert L.mi355_shard_begin(h, t.data_ptr(), len(data), 0, len(data), 0, len(data), C.byref(o), None, C.byref(sh)) == 0
        L.mi355_shard_end(sh)
        assert L.mi355_debug_sort_ranks(h) == 0
    finally:
        L.mi355_deflate_ctx_destroy(h)


# a never-flushed stream keeps a range pending between write() calls -- tokens, workspace and staged input
# of that range -- while the context it was made on (often the shared default one) runs whatever else the process
# encodes.  The ranges therefore live on contexts of the stream's own: two long streams and one-shot calls interleaved
# on ONE context, noise in the data so that Stored blocks cross the seams (their bytes are read from the staged input
# at pack time), and checksum() asked for while a range is pending.
def test_long_streams_interleave_on_one_context(da, ctx, small_ranges):
    import io
    a = datagen.rng_bytes(9_000_000, 0xA1) + datagen.text_like(30_000_000, 0xA2) + datagen.rng_bytes(21_000_000, 0xA3)
    b = datagen.text_like(25_000_000, 0xA4) + datagen.rng_bytes(30_000_000, 0xA5)
    small = datagen.text_like(3_000_000, 0xA6)
    big = datagen.mixed(24_000_000, 0xA7)  # (a one-shot call that goes through ranges itself)
    want_small = ob.encode(small, level=ob.DEFAULT)
    want_big = ob.encode(big, level=ob.DEFAULT)
    ea = da.ZlibEncoder(io.BytesIO(), da.Compression.Default, ctx)
    eb = da.GzEncoder(io.BytesIO(), da.Compression.Default, ctx)
    ra = ob.Stream(ob.make_opts(128, 32, 1, 1))
    rb = ob.Stream(ob.make_opts(128, 32, 1, 2))
    rb.gzip_header(da.BLANK_GZIP_HEADER)
    pa = pb = 0
    step = 6_000_000
    k = 0
    while pa < len(a) or pb < len(b):
        if pa < len(a):
            ea.write_all(a[pa:pa + step])
            ra.write_all(a[pa:pa + step])
            pa += step
        assert ctx.encode(small, da.Compression.Default) == want_small
        if pb < len(b):
            eb.write_all(b[pb:pb + step + 1])
            rb.write_all(b[pb:pb + step + 1])
            pb += step + 1
        if k % 3 == 1:
            assert ea.checksum() == ra.checksum()
            assert eb.checksum() == rb.checksum()
        if k == 4:
            assert ctx.encode(big, da.Compression.Default) == want_big
        k += 1
    assert ea.finish().getvalue() == ra.finish()
    assert eb.finish().getvalue() == rb.finish()


# a slice of tools/fuzz_shard.py: random data kind, size, level and 2-8 virtual ranks (also ranges shorter than a
# block) through the stream-exact sharded encode, against the oracle
def test_randomized_shards(da):
    import random
    import shard
    ctxs = [da.Context(0) for _ in range(8)]
    tally = {}
    for seed in range(1, 41):
        rnd = random.Random(seed * 7919)
        world = rnd.choice([2, 3, 4, 5, 8])
        kind = rnd.choice(["text", "mixed", "rng", "zeros", "period"])
        n = rnd.randrange(world * 140000, world * 140000 + 3_000_000)  # every rank needs more than its halo
        s2 = rnd.randrange(1 << 30)
        data = {"text": lambda: datagen.text_like(n, s2), "mixed": lambda: datagen.mixed(n, s2), "rng": lambda: datagen.rng_bytes(n, s2),
                "zeros": lambda: bytes(n),
                "period": lambda: (datagen.rng_bytes(rnd.choice([3, 300, 4099, 32769]), s2) * (n // 3 + 1))[:n]}[kind]()
        c, l, m = rnd.choice([(1, 0, 0), (128, 32, 1), (128, 32, 1), (0, 0, 1), (0, 0, 0), (32, 8, 1), (500, 64, 1)])
        try:
            ref = ob.encode(data, opts=ob.make_opts(c, l, m))
        except ob.RefPanic:
            tally["ref-panic"] = tally.get("ref-panic", 0) + 1
            continue
        got = shard.encode_p1_virtual(da, ctxs[:world], data, da.CompressionOptions(c, l, m), compat=1)
        assert got == ref, (seed, world, kind, n, (c, l, m))
        tally["ok"] = tally.get("ok", 0) + 1
    assert tally.get("ok", 0) >= 36, tally
    for cx in ctxs:
        cx.close()


# the committed digests of the oracle's streams (tests/golden/oracle_digests.json): every reference fixture,
# every level, raw / zlib / gzip
def test_golden_digests(da, ctx):
    import hashlib
    import json
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import gen_digests
    gold = json.load(open(os.path.join(HERE, "golden", "oracle_digests.json")))
    seen = 0
    for name, data in gen_digests.cases():
        for lvl, (c, l, m) in gen_digests.LV.items():
            for wname, w in (("raw", 0), ("zlib", 1), ("gzip", 2)):
                key = "%s|%s|%s" % (name, lvl, wname)
                if key not in gold["digests"]:
                    continue
                o = da.CompressionOptions(c, l, m)
                z = ctx.encode_gzip(data, o, compat=1) if w == 2 else ctx.encode(data, o, wrapper=w, compat=1)
                assert [len(z), hashlib.sha256(z).hexdigest()] == gold["digests"][key], key
                seen += 1
    assert seen == 390


# error behaviour of the C ABI (include/mi355_deflate.h): codes instead of panics, nothing written past a
# buffer that is too small, handles that refuse use after finish
def test_abi_error_paths(da, ctx):
    import ctypes as C
    L = da.load()
    o = da.CompressionOptions.default().to_c(0, 0, 0)
    data = datagen.text_like(50000, 3)
    n = C.c_size_t(0)
    # output buffer too small: the needed size comes back, the guard bytes stay untouched
    small = (C.c_uint8 * 64)(*([0xAB] * 64))
    rc = L.mi355_deflate_encode(ctx._h, data, len(data), C.byref(o), small, 16, C.byref(n))
    assert rc == da.E_OUT_TOO_SMALL and n.value > 16
    assert bytes(small[16:]) == b"\xAB" * 48
    need = n.value
    big = (C.c_uint8 * need)()
    assert L.mi355_deflate_encode(ctx._h, data, len(data), C.byref(o), big, need, C.byref(n)) == da.OK
    assert bytes(big[:n.value]) == ob.encode(data, level=ob.DEFAULT)
    # bad arguments
    assert L.mi355_deflate_encode(ctx._h, data, len(data), None, big, need, C.byref(n)) == da.E_ARG
    assert L.mi355_deflate_encode(ctx._h, None, 5, C.byref(o), big, need, C.byref(n)) == da.E_ARG
    bad = da.CompressionOptions.default().to_c(3, 0, 0)  # wrapper 3 does not exist
    assert L.mi355_deflate_encode(ctx._h, data, len(data), C.byref(bad), big, need, C.byref(n)) == da.E_ARG
    bad = da.CompressionOptions.default().to_c(0, 0, 2)  # flush mode 2 does not exist
    assert L.mi355_deflate_encode(ctx._h, data, len(data), C.byref(bad), big, need, C.byref(n)) == da.E_ARG
    assert L.mi355_deflate_encode_gzip(ctx._h, data, len(data), C.byref(o), None, 0, big, need, C.byref(n)) == da.E_ARG
    # a device output pointer must be 4-byte aligned
    import torch
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    out = torch.empty(da.bound(len(data)) + 64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    rc = L.mi355_deflate_encode_device(ctx._h, C.c_void_p(t.data_ptr()), len(data), C.byref(o),
                                       C.c_void_p(out.data_ptr() + 1), out.numel() - 1, C.byref(n), None)
    assert rc == da.E_ARG
    # streams: use after finish, gzip header on a non-gzip stream or after data
    h = C.c_void_p()
    assert L.mi355_deflate_stream_new(ctx._h, C.byref(o), C.byref(h)) == da.OK
    assert L.mi355_deflate_stream_gzip_header(h, b"\x1f\x8b", 2) == da.E_STATE
    assert L.mi355_deflate_stream_write(h, data, 100) == da.OK
    assert L.mi355_deflate_stream_finish(h) == da.OK
    assert L.mi355_deflate_stream_write(h, data, 100) == da.E_STATE
    assert L.mi355_deflate_stream_flush(h) == da.E_STATE
    assert L.mi355_deflate_stream_finish(h) == da.E_STATE
    assert L.mi355_deflate_stream_reset(h, None, None) == da.E_STATE
    L.mi355_deflate_stream_free(h)
    g = da.CompressionOptions.default().to_c(2, 0, 0)
    assert L.mi355_deflate_stream_new(ctx._h, C.byref(g), C.byref(h)) == da.OK
    assert L.mi355_deflate_stream_write(h, data, 100) == da.OK
    assert L.mi355_deflate_stream_gzip_header(h, b"\x1f\x8b", 2) == da.E_STATE  # the header goes out with the first write
    L.mi355_deflate_stream_free(h)
    # the context still works after all of that
    assert ctx.encode(data) == ob.encode(data, level=ob.DEFAULT)


# two contexts used from two threads at once (the reference's encoders are independent objects; here one
# context = one HIP stream + workspace, no shared mutable state between contexts)
def test_two_contexts_two_threads(da):
    import threading
    datas = [datagen.text_like(700000, 21), datagen.mixed(900000, 22)]
    refs = [ob.encode(d, level=ob.DEFAULT) for d in datas]
    errs = []

    def work(i):
        c = da.Context(0)
        try:
            for _ in range(15):
                if c.encode(datas[i]) != refs[i]:
                    errs.append(i)
        finally:
            c.close()

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs
    # ... and with pageable buffers large enough for each context's own host threads and rings (deflate_bounce.inc): two sets of
    # threads at work at once, calls of different sizes back to back (the rings wrap, sessions open and close)
    big = [datagen.text_like(23_000_000, 23), datagen.text_like(9_000_000, 24) + datagen.rng_bytes(2_000_000, 25)]
    bigrefs = [ob.encode(d, level=ob.DEFAULT) for d in big]

    def work2(i):
        c = da.Context(0)
        try:
            for k in range(6):
                d = big[(i + k) % 2]
                if c.encode(d) != bigrefs[(i + k) % 2]:
                    errs.append(("pageable", i, k))
                cut = 4_500_000 + 333_333 * k
                if c.encode(d[:cut]) != ob.encode(d[:cut], level=ob.DEFAULT):
                    errs.append(("pageable cut", i, k))
        finally:
            c.close()

    ts = [threading.Thread(target=work2, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs


# the plain-C example over the ABI (examples/mi355_deflate_cli.c), built with gcc and run as a process
def test_c_example_program(tmp_path):
    import subprocess
    exe = str(tmp_path / "cli")
    subprocess.run(["gcc", "-O2", "-std=c99", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "mi355_deflate_cli.c"), "-L", os.path.join(ROOT, "deflate-rs_amd"),
                    "-lmi355deflate", "-Wl,-rpath," + os.path.join(ROOT, "deflate-rs_amd"), "-o", exe], check=True)
    trap = str(tmp_path / "segv_trap.so")
    subprocess.run(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", trap, os.path.join(ROOT, "tools", "probes", "segv_trap.c"), "-ldl"],
                   check=True)
    env = dict(os.environ, LD_PRELOAD=trap)
    src = os.path.join(FIX, "pg11.txt")
    data = open(src, "rb").read()
    blank = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff])
    for flag, lvl, expect in (("-raw", ("-default", ob.DEFAULT), None), ("-zlib", ("-best", ob.BEST), None),
                              ("-gzip", ("-fast", ob.FAST), None)):
        for extra in ([], ["-chunk", "5000"]):
            out = str(tmp_path / "out.bin")
            cmd = [exe, flag, lvl[0]] + extra + [src, out]
            # (no retry: round 2 saw one SIGSEGV at process exit in ~170 runs and hid it behind one; 2 100 runs under
            # the fault trap in round 3 -- alone, four at a time, under a parent that holds a context -- had none,
            # DESIGN.md.  The trap stays loaded so that a crash, should it come back, arrives with its backtrace.)
            r = subprocess.run(cmd, capture_output=True, env=env)
            assert r.returncode == 0, (cmd, r.returncode, r.stderr[-3000:])
            got = open(out, "rb").read()
            if flag == "-gzip":
                want = ob.encode_gzip(data, blank, level=lvl[1])
            else:
                want = ob.encode(data, level=lvl[1], wrapper=1 if flag == "-zlib" else 0)
            assert got == want, (flag, lvl[0], extra)


# The restart steps (lz77.rs:305-547 seen from a position) worked out by the kernel that writes the tokens, from the match table --
# the lazy step as a run of per-position bits, a step read off its filed entry -- or by k_adv into device memory
# (MI355_CFG_STEPS_IN_EMIT 1 / 0): the oracle's bytes either way, on inputs with long deferral chains, ends inside a chain,
# segments that end in the middle of a match, and the levels that take the one way or the other.
def test_restart_steps_in_the_token_kernel_or_from_k_adv(da):
    ctx = da.Context(0)
    try:
        rnd = __import__("random").Random(77)
        # ascending matches in a row: position p+1 has a longer match than p, many times over (chains of deferrals)
        base = bytes(rnd.randrange(256) for _ in range(600))
        climb = b"".join(base[i:i + 40 + k] + bytes([k & 0xFF, (k * 7) & 0xFF]) for k, i in enumerate(range(0, 400, 3)))
        inputs = [("text", datagen.text_like(3_000_001, 31)), ("mixed", datagen.mixed(2_000_000, 32)),
                  ("chains", (base + climb) * 40), ("short", datagen.text_like(1151, 33)), ("two", b"ab"),
                  ("tail", datagen.text_like(70_000, 34) + b"xyzxyzxyzxy")]
        for name, data in inputs:
            for lv in ("default", "fast", "best"):
                for where in (1, 0):
                    ctx.config(da.Context.CFG_STEPS_IN_EMIT, where)
                    try:
                        agree(da, ctx, data, *LV[lv])
                    except AssertionError as e:
                        raise AssertionError("%s %s MI355_CFG_STEPS_IN_EMIT=%d: %s" % (name, lv, where, e))
    finally:
        ctx.close()
