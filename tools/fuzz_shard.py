"""GPU box: randomized check of the stream-exact sharded encode (virtual ranks on one GPU) against the
oracle.  usage: fuzz_shard.py [cases] [first_seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: F401
import datagen, deflate_amd as da, oracle_binding as ob, shard, fuzz_gpu

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctxs = [da.Context(0) for _ in range(8)]
bad = 0
tally = {}
for seed in range(first, first + cases):
    rnd = random.Random(seed * 7919)
    world = rnd.choice([2, 3, 4, 5, 8])
    kind = rnd.choice(["text", "mixed", "rng", "zeros", "period"])
    n = rnd.randrange(world * 140000, world * 140000 + 3_000_000)   # every rank needs more than its halo
    s2 = rnd.randrange(1 << 30)
    data = {"text": lambda: datagen.text_like(n, s2), "mixed": lambda: datagen.mixed(n, s2), "rng": lambda: datagen.rng_bytes(n, s2),
            "zeros": lambda: bytes(n), "period": lambda: (datagen.rng_bytes(rnd.choice([3, 300, 4099, 32769]), s2) * (n // 3 + 1))[:n]}[kind]()
    c, l, m = rnd.choice([(1, 0, 0), (128, 32, 1), (128, 32, 1), (0, 0, 1), (0, 0, 0), (32, 8, 1), (500, 64, 1)])
    try:
        ref = ob.encode(data, opts=ob.make_opts(c, l, m))
        got = shard.encode_p1_virtual(da, ctxs[:world], data, da.CompressionOptions(c, l, m), compat=1)
        r = "ok" if got == ref else "DIFF"
    except ob.RefPanic:
        r = "ref-panic"
    except ValueError as e:   # a rank with fewer tokens than a block boundary needs (tiny, very compressible shards)
        r = "refused"
    tally[r] = tally.get(r, 0) + 1
    if r == "DIFF":
        bad += 1
        print("seed", seed, "DIFF world", world, kind, n, (c, l, m), flush=True)
print("fuzz_shard:", tally)
sys.exit(1 if bad else 0)
