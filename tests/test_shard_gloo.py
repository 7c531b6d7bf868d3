"""N > 1 path on CPU: two gloo ranks shard one input, encode their chunks (here with the oracle's
streaming encoder standing in for the per-rank GPU encode, since this container has no GPU) and
stitch them on rank 0 exactly as bench.py / shard.py do over RCCL.  The stitched stream must be the
chunk-exact (P2) stream and inflate to the input."""
import os
import sys
import zlib

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "deflate-rs_amd"))
sys.path.insert(0, HERE)


def _worker(rank, world, port, path, q):
    import oracle_binding as ob
    import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = open(path, "rb").read()
    lo, hi = shard.shard_range(len(data), rank, world)
    s = ob.Stream(ob.preset(ob.DEFAULT))
    s.write_all(data[lo:hi])
    if shard.flush_mode_for(rank, world) == 1:
        s.flush()
        chunk = s.output()
    else:
        chunk = s.finish()
    t = torch.frombuffer(bytearray(chunk) + bytearray(16), dtype=torch.uint8)
    buf, total = shard.stitch(t, len(chunk), rank, world)
    if rank == 0:
        out = bytes(buf.numpy())
        d = zlib.decompressobj(-15)
        ok = (d.decompress(out) + d.flush()) == data and d.eof and len(out) == total
        q.put((ok, len(out), len(ob.encode(data, level=ob.DEFAULT))))
    dist.barrier()
    dist.destroy_process_group()


def _run(world):
    path = os.path.join(HERE, "golden", "ref_inputs", "pg11.txt")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, path, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok, n, n_p1 = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok
    # chunking costs a few bytes per seam plus the lost cross-chunk history
    assert n_p1 <= n < n_p1 * 1.1


def test_two_rank_stitch():
    _run(2)


def test_three_rank_stitch():
    _run(3)


def test_shard_ranges_cover_input():
    import shard
    for total in (0, 1, 100, 32768, 100000, 10 ** 8 + 7):
        for world in (1, 2, 3, 8):
            rs = [shard.shard_range(total, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))


def test_p1_exchange_math():
    import shard
    # token split: every non-final rank ends on a block boundary after the exchange
    counts = [100000, 70000, 40000, 50000]
    skip, tail = shard.p1_token_split(counts)
    first = [0, 100000, 170000, 210000]
    for r in range(4):
        assert (first[r] + skip[r]) % shard.BLOCK_TOKENS == 0
        owned = counts[r] - skip[r] + tail[r]
        if r < 3:
            assert owned % shard.BLOCK_TOKENS == 0
    assert tail[:3] == skip[1:]
    # a range that does not reach its next block boundary owns no block: all of its tokens go left, and the
    # block they belong to takes the rest of what it needs from the rank after it
    skip, tail, owns, pieces = shard.p1_token_plan([100000, 5, 100000])
    B = shard.BLOCK_TOKENS
    assert skip[1] == 5 and tail[1] == 0 and not owns[1]
    assert tail[0] == (-100000) % B and pieces[0] == [(1, 5), (2, tail[0] - 5)]
    assert skip[2] == tail[0] - 5 and owns == [False, False, True]
    import random
    rnd = random.Random(5)
    for _ in range(3000):  # whole blocks everywhere but at the one owner of the last block; every block owned once
        counts = [rnd.choice([0, 1, 7, B - 1, B, B + 1, rnd.randrange(0, 3 * B), rnd.randrange(0, 300)])
                  for _ in range(rnd.randrange(1, 9))]
        skip, tail, owns, pieces = shard.p1_token_plan(counts)
        assert sum(owns) == 1
        blocks = 0
        for r, c in enumerate(counts):
            own = c - skip[r] + tail[r]
            assert sum(k for _, k in pieces[r]) == tail[r] and all(q > r for q, _ in pieces[r])
            assert owns[r] or own % B == 0
            blocks += own // B + (1 if owns[r] else 0)
        assert blocks == sum(counts) // B + 1
    # entries: a rank whose range is jumped over keeps the incoming position
    lays = [dict(a=0, b=1000), dict(a=1000, b=1200), dict(a=1200, b=3000)]
    tabs = [[300] * shard.ZONE, [7] * shard.ZONE, [0] * shard.ZONE]
    assert shard.p1_entries(lays, tabs) == [0, 1300, 1300]
    # entries by speculation: every rank reports (held, entry, exit) in its own buffer coordinates; the chain holds iff rank 0
    # was entered at 0 and every other rank where the rank before it was left -- anything else hands over to the exit tables
    lays = [dict(a=0, b=65536, g_lo=0), dict(a=65536, b=131072, g_lo=32768), dict(a=131072, b=200000, g_lo=98304)]
    good = [(True, 0, 65540), (True, 65540 - 32768, 131075 - 32768), (True, 131075 - 98304, 200000 - 98304)]
    assert shard.p1_spec_entries(lays, good) == [0, 65540, 131075]
    assert shard.p1_spec_entries(lays, [good[0], (False, 0, 0), good[2]]) is None          # a rank's own chain failed
    assert shard.p1_spec_entries(lays, [good[0], (True, 65541 - 32768, good[1][2]), good[2]]) is None  # entered elsewhere
    assert shard.p1_spec_entries(lays, [(True, 3, 65540), good[1], good[2]]) is None       # rank 0 not entered at 0
    # (a range jumped over: entered and left at the same position behind its end)
    assert shard.p1_spec_entries(lays[:2], [(True, 0, 131080), (True, 131080 - 32768, 131080 - 32768)]) == [0, 131080]
    # layouts cover the input with history and look-ahead clipped to it
    for total in (5_000_000, 40_000_001):
        for world in (1, 3, 8):
            ls = [shard.p1_layout(total, r, world) for r in range(world)]
            assert ls[0]["g_lo"] == 0 and ls[-1]["g_hi"] == total
            for L in ls:
                assert L["g_lo"] % 32768 == 0 and L["lo"] % 1024 == 0
