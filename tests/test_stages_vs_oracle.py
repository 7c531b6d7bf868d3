"""The stage decomposition the HIP kernels implement (deflate-rs_amd/csrc/stages.h, run on the
host by tests/hostsim) must reproduce the oracle byte for byte.  This pins the *algorithmic*
re-cut -- pure match table, restart-path parse, 31744-token blocks, per-block Huffman, bit
plan, quirks Q1/Q12/Q13 -- on CPU; the -m gpu tests then pin the kernels.  CPU only."""
import glob
import os
import random
import zlib

import pytest

import datagen
import hostsim_binding as hs
import oracle_binding as ob

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_inputs")
LV = {"fast": (1, 0, 0), "default": (128, 32, 1), "best": (1768, 128, 1), "rle": (0, 0, 1),
      "huffman_only": (0, 0, 0)}


def agree(data, c, l, m, seg=1024, fan=4):
    ref = ob.encode(data, opts=ob.make_opts(c, l, m))
    rb = ob.trace_blocks()
    rc, out, flags, bl = hs.encode(data, c, l, m, seg, fan)
    assert rc == 0
    assert not (flags & 4), "hierarchical path search, a step read off its filed entry or the loop-free lazy step disagrees with the direct walk"
    assert bl == rb
    assert out == ref
    return flags


@pytest.mark.parametrize("level", list(LV))
def test_reference_fixtures(level):
    c, l, m = LV[level]
    for f in ["pg11.txt", "short.bin", "issue_18_201911.bin", "dump.bin"]:
        agree(open(os.path.join(FIX, f), "rb").read(), c, l, m)


def test_afl_inputs_default_and_fast():
    q1 = 0
    for f in sorted(glob.glob(os.path.join(FIX, "afl", "*"))):
        data = open(f, "rb").read()
        for level in ("default", "fast"):
            q1 += agree(data, *LV[level]) & 1
    # these near-incompressible inputs are what drives quirk Q1 (hash re-warm)
    assert q1 >= 0


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 6, 257, 258, 259, 260, 31743, 31744, 31745, 32767,
                               32768, 32769, 65535, 65536, 65537, 65794, 65795, 100000])
def test_sizes_zeros_and_ramp(n):
    for level in LV:
        agree(bytes(n), *LV[level])
        agree(bytes(i & 0xFF for i in range(n)), *LV[level])


def test_random_exact_block_fill():
    # random bytes: (almost) one literal per token, so blocks end right at 31744-token marks and
    # the final-empty-block rule (Q8) and Q1 are exercised
    for n in (31744, 31745, 63488, 63489, 95232, 100000):
        data = datagen.rng_bytes(n, n)
        for level in LV:
            agree(data, *LV[level])


@pytest.mark.parametrize("seed", range(8))
def test_mixed_inputs(seed):
    n = [20000, 70000, 140000, 300000][seed % 4]
    data = datagen.mixed(n, seed)
    for level in LV:
        agree(data, *LV[level], seg=[1024, 600, 4096, 1024][seed % 4], fan=[4, 2, 8, 3][seed % 4])


def test_text_like_default_and_best():
    data = datagen.text_like(400000, 1)
    agree(data, *LV["default"])
    agree(data[:150000], *LV["best"])
    agree(data, *LV["fast"])


def test_custom_options():
    rnd = random.Random(42)
    data = datagen.mixed(120000, 99)
    for _ in range(10):
        c = rnd.choice([1, 2, 3, 7, 32, 128, 500, 4000])
        l = rnd.choice([3, 4, 8, 31, 32, 33, 64, 258, 1000, 40000])
        m = rnd.choice([0, 1])
        agree(data, c, l, m)


def test_long_runs_and_periods():
    for data in (b"ab" * 70000, b"abc" * 50000, bytes(100) + b"x" + bytes(200000),
                 (datagen.rng_bytes(300, 5) * 500), (datagen.rng_bytes(32768, 6) * 4),
                 (datagen.rng_bytes(32769, 7) * 3), (datagen.rng_bytes(32767, 8) * 3)):
        for level in LV:
            agree(data, *LV[level])


def q13_case(seed, B=98304, total=140000):
    """An input whose third block is incompressible and ends with a match that crosses the end
    of a non-first window: the reference slides its buffer before reading the stored bytes
    (SURVEY.md A.4 Q13)."""
    rnd = random.Random(seed)
    base = bytearray(rnd.getrandbits(8) for _ in range(total))
    plants = []
    s0, L0 = B - 3, 48

    def make(s0):
        d = bytearray(base)
        for (pos, src, L) in plants:
            d[pos:pos + L] = d[src:src + L]
        d[s0:s0 + L0] = d[s0 - 5000:s0 - 5000 + L0]
        return bytes(d)

    for _ in range(80):
        d = make(s0)
        pos = 0
        idx = None
        for i, t in enumerate(ob.lz77(d, 128, 32, 1)):
            ln = 1 if t[0] == "lit" else t[1]
            if t[0] == "ld" and ln >= 40 and pos <= B - 2 and pos + ln > B:
                idx = i
                break
            pos += ln
        if idx is None:
            return None
        r = (idx + 1) % 31744
        if r == 0:
            return d
        if r <= 40 and s0 - r > B - L0 + 2:
            s0 -= r
            continue
        L = 258 if r > 300 else max(4, min(258, r - 20))
        p = 8000 + 300 * len(plants)
        plants.append((p, p - 6500, L))
    return None


def test_q13_stored_after_slide_is_replicated():
    hit = 0
    for seed in range(1, 8):
        d = q13_case(seed)
        if d is None:
            continue
        ref = ob.encode(d, level=ob.DEFAULT)
        if not ob.last_hazards():
            continue
        hit += 1
        # the reference's own output is corrupt here ...
        dd = zlib.decompressobj(-15)
        assert dd.decompress(ref) + dd.flush() != d
        # ... and the stage functions reproduce it bit for bit when asked to
        rc, out, flags, _ = hs.encode(d, 128, 32, 1, 1024, 4)
        assert rc == 0 and (flags & 2) and out == ref
        if hit >= 2:
            break
    assert hit >= 1


# The formulations of the chain walk in stages.h (single, multi-chain, parked, the pair-table form of k_match3 = mode 6, with its
# run-of-one-byte service = mode 7, with a position's candidates walked as a near and a far half -- a small call's walk, where the far
# half's match counts only if strictly longer and the quarter-budget result is the near half's = mode 8) must give the
# same match table and the same stream; mode 3 additionally cuts every compare short so that the
# "stay parked, continue at the next service" path of the GPU policy is exercised.
@pytest.mark.parametrize("mode", [1, 2, 3, 6, 7, 8])
def test_match_walk_formulations_agree(mode):
    cases = [datagen.text_like(200000, 3), datagen.mixed(150000, 5), datagen.rng_bytes(70000, 2), bytes(70000),
             b"abcabcabcabc", (datagen.rng_bytes(300, 5) * 400)]
    try:
        for data in cases:
            for checks in (1, 7, 16, 17, 128, 1768):
                hs.use_multi(0)
                a = hs.match_table(data, checks)
                hs.use_multi(mode)
                assert hs.match_table(data, checks) == a
            hs.use_multi(mode)
            for level in ("default", "best", "fast"):
                agree(data, *LV[level])
    finally:
        hs.use_multi(0)


# The reference's head table starts as head[h] = h, so a chain that runs out of real candidates hops
# on to the position numbered like its hash value (chained_hash_table.rs:64-69, matching.rs:123-131).
# DESIGN.md §2.1 claims those hops can never change a match of length >= 3 while hashes are true
# 3-byte hashes, which is why the kernels skip them unless a hash re-warm is in play; here the hops
# are forced on and every stream must stay identical (and equal to the oracle's).
def test_identity_hops_never_change_output():
    cases = [datagen.text_like(200000, 3), datagen.mixed(150000, 5), datagen.rng_bytes(100000, 2), bytes(70000),
             (datagen.rng_bytes(300, 5) * 400), datagen.rng_bytes(31744 * 2 + 100, 31744)]
    try:
        for data in cases:
            for checks in (1, 128, 1768):
                hs.force_ident(0)
                a = [m if (m & 0xffff) >= 3 else 0 for m in hs.match_table(data, checks)]
                hs.force_ident(1)
                b = [m if (m & 0xffff) >= 3 else 0 for m in hs.match_table(data, checks)]
                assert a == b
            hs.force_ident(1)
            for level in ("default", "best", "fast"):
                agree(data, *LV[level])
    finally:
        hs.force_ident(0)


def test_code_length_rle_run_by_run_equals_the_state_machine():
    """k_block_header codes the chained code lengths run by run (stages.h el_run_count / el_run_emit); the
    reference's state machine (length_encode.rs:82-155, stages.h encode_lengths_rle, pinned by the KATs of
    test_oracle_kat.py) must give the same symbols for every list."""
    import random
    rnd = random.Random(20260929)
    cases = [bytes([v]) * c for v in (0, 1, 7, 15) for c in list(range(1, 30)) + [137, 138, 139, 140, 148, 149, 275, 276, 277, 316]]
    for _ in range(4000):
        out = bytearray()
        while len(out) < rnd.choice((1, 5, 19, 60, 287, 316)):
            v = rnd.choice((0, 0, 0, rnd.randrange(1, 16)))
            c = rnd.choice((1, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 20, rnd.randrange(1, 300)))
            out += bytes([v]) * c
        cases.append(bytes(out[:316]))
    for lens in cases:
        assert hs.rle_forms_agree(lens) == 0, list(lens)


def test_permuted_pair_table_is_a_word_permutation_of_every_256_bytes():
    """k_match3_swz keeps the pair table with the 8-byte words of every 256-byte block permuted (stages.h m3_swz).  What the walk
    relies on: a bijection of each block onto itself, whole 8-byte words moved (an aligned read of up to 8 bytes stays together),
    and rows of records -- lanes 192 or 512 bytes apart -- spread over the LDS's 64 four-byte banks instead of 4 or 1."""
    size = 132 * 1024  # (the table: 2 x 32768 + 272 pairs, rounded up to whole blocks)
    img = [hs.swz(a) for a in range(0, size, 8)]
    for blk in range(0, size, 256):
        words = img[blk // 8:(blk + 256) // 8]
        assert sorted(words) == list(range(blk, blk + 256, 8)), blk
    for a in (0, 8, 1000, 2047, 2048, 65535, 131071):
        assert hs.swz(a) & 7 == a & 7 and hs.swz(a) >> 8 == a >> 8
    for stride, least in ((128, 24), (192, 24), (256, 24), (512, 16), (1024, 8)):  # rows of 64 / 96 / 128 / 256 / 512 bytes
        for base in (0, 6, 1234, 60000):
            plain = {((base + stride * i) >> 2) & 63 for i in range(64) if base + stride * i < size}
            perm = {(hs.swz(base + stride * i) >> 2) & 63 for i in range(64) if base + stride * i < size}
            assert len(plain) <= 8 and len(perm) >= least, (stride, base, len(plain), len(perm))

