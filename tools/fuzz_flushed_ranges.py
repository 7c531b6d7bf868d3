"""GPU box: differential test of streams whose chunks between flushes are long enough to be handed over in ranges
(16 MiB ranges): random flush points -- inside the first three windows (such a chunk is one pass), at and around window
edges, anywhere -- with 1-byte, 2-byte and longer first writes behind them and flushes a few bytes apart, every level and
framing, against the oracle driven with the same calls.
usage: fuzz_flushed_ranges.py [cases] [first_seed] [early]   (early: every case has a chunk of 40 MB or more right behind a flush
point inside the first windows of the stream, where Q1 and the hash re-warm at flush points apply)"""
import io, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import datagen, deflate_amd as da, oracle_binding as ob

LV = {"fast": (1, 0, 0), "default": (128, 32, 1), "best": (1768, 128, 1), "rle": (0, 0, 1)}


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    early = len(sys.argv) > 3  # every case: a long chunk right behind a flush point inside the first windows
    ctx = da.Context(0)
    ctx.config(ctx.CFG_RANGE_BYTES, 16 << 20)
    L = da.load()
    pool = datagen.text_like(50_000_000, 0xF1) + datagen.mixed(30_000_000, 0xF2) + bytes(6_000_000) + datagen.rng_bytes(9_000_000, 0xF3)
    ok = 0
    for seed in range(first, first + cases):
        rnd = random.Random(seed)
        n = rnd.randrange(60_000_000, len(pool))
        off = rnd.randrange(0, len(pool) - n + 1)
        data = pool[off:off + n]
        if rnd.random() < 0.45:  # noise at the stream's start: a block that fills inside the first window (Q1), stored blocks
            k = rnd.choice([20_000, 40_000, 70_000, 200_000])
            data = datagen.rng_bytes(k, seed) + data[k:]
        edge = rnd.choice([0, 1, 2, 32767, 32768, 32769, 98303, 98304, 98305, 12345])
        if early:
            first_pt = rnd.choice([1, 2, 3, 5, 300, 20_000, 31_000, 32_767, 32_768, 33_000, 40_000, 65_535, 65_536, 65_537, 70_000, 90_000, 98_303])
            points = [first_pt]
        else:
            points = [rnd.choice([1, 2, 5, 20_000, 31_000, 33_000, 40_000, 65_536, 65_537, 90_000, 98_304 + edge,
                                  rnd.randrange(200_000, 20_000_000) // 32768 * 32768 + edge]),
                      rnd.randrange(1_000_000, n - 40_000_000)]
        if rnd.random() < 0.5:
            points.append(rnd.randrange(1_000_000, n))
        points = sorted(set(points))
        if early:  # an early flush point (or two of them close together) and nothing else before a chunk of at least 40 MB
            points = [points[0]] + [p for p in points[1:] if p > points[0] + 40_000_000]
        if rnd.random() < 0.5:  # a second flush a few bytes behind one of them
            q = rnd.choice(points)
            points = sorted(set(points + [q + rnd.choice([1, 2, 3, 700])]))
        lv = rnd.choice(list(LV))
        c, l, m = LV[lv]
        wrapper = rnd.choice([0, 1, 2])
        cls = (da.DeflateEncoder, da.ZlibEncoder, da.GzEncoder)[wrapper]
        enc = cls(io.BytesIO(), da.CompressionOptions(c, l, m), ctx)
        ref = ob.Stream(ob.make_opts(c, l, m, wrapper))
        if wrapper == 2:
            ref.gzip_header(da.BLANK_GZIP_HEADER)
        pos, held, after, todo = 0, 0, False, [p for p in points if p < n]
        while pos < n:
            step = rnd.choice([1, 2, 3, 70_000]) if after else rnd.choice([4000, 65_536, 1_000_003, 4_500_000])
            after = False
            if todo and pos < todo[0] <= pos + step:
                step = todo[0] - pos
            step = min(step, n - pos)
            enc.write_all(data[pos:pos + step]); ref.write_all(data[pos:pos + step]); pos += step
            held = max(held, L.mi355_deflate_stream_held_bytes(enc._s))
            if todo and pos == todo[0]:
                todo.pop(0)
                enc.flush(); ref.flush()
                after = True
        got = enc.finish().getvalue()
        want = ref.finish()
        if got != want:
            print("DIFF seed", seed, lv, wrapper, points, n, off, len(got), len(want))
            sys.exit(1)
        ok += 1
        print("seed %d ok: %s wrapper %d, %d bytes, flushes at %s, handle held at most %.1f MB" % (seed, lv, wrapper, n, points, held / 1e6), flush=True)
    print("%d cases, none different" % ok)


main()
