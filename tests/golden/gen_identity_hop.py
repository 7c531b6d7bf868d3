"""Builds tests/golden/identity_hop.bin: an input on which a chain of the reference leaves its bucket through an identity entry of
the head table AFTER the first slide and finds a real match there -- the case the sorted walk of the GPU path has no array for.

How the reference gets there (all of it literal behaviour of /root/reference, restated in oracle/deflref.cpp):
  * the first block of the stream fills (31 744 LZ values, output_writer.rs:19) exactly at the end of the first window: 31 232
    literals and 512 matches of three bytes cover 32 768 bytes.  The block ends inside the first window with nothing pending, so the
    next call of lz77_compress_block warms the rolling hash up again with data[0], data[1] (quirk Q1, lz77.rs:628-638) and the next two
    positions that are filed -- 32 768 and 32 769 -- land in chains that have nothing to do with their bytes;
  * the three bytes at 32 768 hash to 0 and those at 32 769 hash to 1 (chained_hash_table.rs:55-62), and no other position of the
    second window is filed under 0 or 1;
  * after the second window the table slides (chained_hash_table.rs:197-219): head[0] and head[1] go back to "the position numbered
    like the hash", i.e. buffer positions 0 and 1 = stream positions 32 768 and 32 769;
  * the third window begins with a copy of the bytes at 32 768: position 65 536 hashes to 0, its chain is the identity entry alone,
    and the candidate there -- exactly 32 768 bytes back, the far end of the window (matching.rs:102-106) -- is a real match.
    Position 65 537 does the same through bucket 1.

The generator checks against the oracle that all of this happens (the token at 65 536 is a match at distance 32 768) and that the
stream changes when the copy at 65 536 is replaced by other bytes; it writes the input and the oracle's digests.
    python tests/golden/gen_identity_hop.py
"""
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tools"))
import oracle_binding as ob  # noqa: E402
import tokdump  # noqa: E402

W = 32768


def hash3(a, b, c):
    return (((a & 31) << 10) ^ (b << 5) ^ c) & 0x7FFF


class Filler:
    """bytes whose trigrams are all different (no match of three bytes anywhere) and never hash to a reserved value"""

    def __init__(self, seed, reserved_hashes):
        self.r = random.Random(seed)
        self.seen = set()
        self.reserved = set(reserved_hashes)
        self.out = bytearray()

    def ok(self, a, b, c):
        return (a, b, c) not in self.seen and hash3(a, b, c) not in self.reserved

    def push_fixed(self, bs):
        for x in bs:
            self.out.append(x)
            if len(self.out) >= 3:
                self.seen.add(tuple(self.out[-3:]))

    def push_random(self, n):
        for _ in range(n):
            for _try in range(1000):
                x = self.r.randrange(256)
                if len(self.out) < 2 or self.ok(self.out[-2], self.out[-1], x):
                    break
            else:
                raise RuntimeError("no byte fits")
            self.push_fixed([x])


def build():
    f = Filler(20260930, reserved_hashes=[0, 1])
    # ---- window 1: 31 232 literals + 512 three-byte matches = 31 744 values over 32 768 bytes ----
    # a match: three bytes copied from 40 positions back, then a byte that differs from the one behind the source
    # 512 groups of 61 new bytes and a copy of three bytes from 40 positions back; the byte behind a copy differs from the
    # one behind its source, so the match is exactly three long
    for _group in range(512):
        start = len(f.out)
        while len(f.out) - start < 61:
            if len(f.out) - start == 0 and start:
                nxt = f.out[start - 3 - 40 + 3]  # the byte behind the last copy's source
                for _try in range(1000):
                    x = f.r.randrange(256)
                    if x != nxt and f.ok(f.out[-2], f.out[-1], x):
                        break
                f.push_fixed([x])
            else:
                f.push_random(1)
        src = len(f.out) - 40
        cp = bytes(f.out[src:src + 3])
        # (the trigrams that straddle the copy's start must be new as well: draw the byte in front of it again until they are)
        for _try in range(1000):
            t1, t2 = (f.out[-2], f.out[-1], cp[0]), (f.out[-1], cp[0], cp[1])
            if t1 not in f.seen and t2 not in f.seen and hash3(*t1) > 1 and hash3(*t2) > 1:
                break
            f.seen.discard(tuple(f.out[-3:]))
            f.out.pop()
            f.push_random(1)
        else:
            raise RuntimeError("no byte fits in front of a copy")
        f.seen.add(t1)
        f.seen.add(t2)
        f.out += cp
    assert len(f.out) == W, len(f.out)
    w1 = bytes(f.out)
    # ---- window 2 begins with Z: trigram of hash 0, then hash 1 at the position behind it ----
    # hash3(a, b, c) = 0 with a & 31 = 0, b = 1, c = 0x20; hash of (b, c, d) = 1: ((1 & 31) << 10) ^ (0x20 << 5) ^ d = 1 -> d = 1
    z = bytearray([0x40, 0x01, 0x20, 0x01])
    assert hash3(*z[0:3]) == 0 and hash3(*z[1:4]) == 1
    g = Filler(77, reserved_hashes=[0, 1])
    g.seen = f.seen
    g.out = bytearray(w1)
    g.reserved = set()        # Z itself holds the two reserved trigrams
    g.push_fixed(z)
    g.reserved = {0, 1}
    g.push_random(60)         # the rest of Z: 64 bytes that occur nowhere else
    z_full = bytes(g.out[W:W + 64])
    g.push_random(W - 64)     # the rest of window 2
    assert len(g.out) == 2 * W
    # ---- window 3: Z again, then filler ----
    g.out += z_full
    g.push_random(1)          # (not through push_fixed: Z's trigrams are "seen"; the byte behind the copy must differ from the one behind Z)
    while g.out[-1] == g.out[W + 64]:
        g.out[-1] = (g.out[-1] + 1) & 0xFF
    g.push_random(3000)
    return bytes(g.out), z_full


def main():
    data, z = build()
    opts = ob.make_opts(128, 32, 1)
    ref = ob.encode(data, opts=opts)
    tr = ob.trace_blocks()
    assert tr[0]["n_lz"] == 31744 and tr[0]["in_bytes"] == W, tr[0]
    blocks = tokdump.tokens(ref)
    toks = [t for b in blocks for t in b["toks"]]
    at = {t[0]: t for t in toks}
    assert 65536 in at and at[65536][1] >= 60 and at[65536][2] == W, at.get(65536)
    # the same input with other bytes where the copy was: another stream (the match is what the identity entry gave)
    other = bytearray(data)
    other[65536] ^= 0x5A
    assert ob.encode(bytes(other), opts=opts) != ref
    out = os.path.join(HERE, "identity_hop.bin")
    open(out, "wb").write(data)
    dig = {}
    for name, (c, l, m) in {"default": (128, 32, 1), "best": (1768, 128, 1), "fast": (1, 0, 0), "greedy128": (128, 0, 0)}.items():
        s = ob.encode(data, opts=ob.make_opts(c, l, m))
        bl = [t for b in tokdump.tokens(s) for t in b["toks"] if t[0] in (65536, 65537)]
        dig[name] = {"len": len(s), "sha256": hashlib.sha256(s).hexdigest(), "tokens_at_65536_65537": bl}
    json.dump({"input_sha256": hashlib.sha256(data).hexdigest(), "input_len": len(data), "streams": dig},
              open(os.path.join(HERE, "identity_hop.json"), "w"), indent=1)
    print("wrote", out, len(data), "bytes; token at 65536:", at[65536], "block 0:", tr[0])
    for k, v in dig.items():
        print(k, v)


if __name__ == "__main__":
    main()
