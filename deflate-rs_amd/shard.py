"""Sharding one input over the GPUs of a node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for the tests).

Two ways, DESIGN.md section 6:

* stream-exact ("P1", the default of bench.py): the ranks together produce the very bytes one
  encoder produces for the whole input.  Every rank runs links / match / parse steps on its own
  byte range (+32 KiB history, +66 KiB look-ahead) and four small exchanges place it in the global
  stream: exit tables (parse entry), token counts (+ the <= 31 743 tokens of a block that straddles
  two ranks), per-block costs (every rank then runs the same serial block plan), and the final
  OR-stitch of the packed byte ranges onto rank 0.  p1_* / encode_p1_* below.
* chunk-exact ("P2"): every rank encodes its range as one chunk, all but the last in the reference's
  sync-flush form (include/mi355_deflate.h MI355_FLUSH_SYNC), and the byte-aligned chunks are
  concatenated on rank 0.  Equal to concat_i fresh_reference_encoder(chunk_i).write_all().flush()
  (finish() for the last); matches do not cross rank boundaries.  shard_range / stitch below.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous, 32 KiB-aligned byte range of `rank` (the last rank takes the remainder)."""
    per = (total // world) // 32768 * 32768
    if per == 0:
        per = total // world
    lo = rank * per
    hi = total if rank == world - 1 else lo + per
    return lo, hi


def flush_mode_for(rank, world):
    return 0 if rank == world - 1 else 1  # MI355_FLUSH_FINISH / MI355_FLUSH_SYNC


def stitch(local_out, local_len, rank, world, group=None):
    """Gather the chunks onto rank 0.  local_out: uint8 tensor (device for nccl, cpu for gloo) whose
    first local_len bytes are this rank's chunk.  Returns (tensor, total_len) on rank 0, (None,
    total_len) elsewhere."""
    if world == 1:
        return local_out[:local_len], local_len
    dev = local_out.device
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local_len], dtype=torch.int64, device=dev), group=group)
    sizes = [int(s.item()) for s in sizes]
    total = sum(sizes)
    if rank == 0:
        buf = torch.empty(total, dtype=torch.uint8, device=dev)
        buf[: sizes[0]] = local_out[: sizes[0]]
        off = sizes[0]
        reqs = []
        for r in range(1, world):
            if sizes[r]:
                reqs.append(dist.irecv(buf[off: off + sizes[r]], src=r, group=group))
            off += sizes[r]
        for q in reqs:
            q.wait()
        return buf, total
    if local_len:
        dist.send(local_out[:local_len].contiguous(), dst=0, group=group)
    return None, total


# =====================================================================================================
# Stream-exact (P1) sharding: the stitched stream equals the reference run on the WHOLE input.
# include/mi355_deflate.h ("sharded encode") describes the per-rank phases; here is what travels
# between them.  All of it is tiny next to the data: 576 x 4 bytes of exit table, 8 bytes of token
# count, at most 31743 x 4 bytes of straddling tokens and 48 bytes per block of costs per rank.
# =====================================================================================================
HISTORY = 32768          # bytes of history in front of a rank's range (the match window)
LOOKAHEAD = 66 * 1024    # bytes behind it: 258 of match look-ahead + room for a stored straddling block
BLOCK_TOKENS = 31744
ZONE = 576


def p1_layout(total, rank, world):
    """-> dict(a, b: the rank's byte range; g_lo, g_hi: the bytes it must hold; lo, hi: a, b in buffer
    coordinates)."""
    a, b = shard_range(total, rank, world)
    g_lo = max(0, a - HISTORY)
    g_hi = min(total, b + LOOKAHEAD)
    return dict(a=a, b=b, g_lo=g_lo, g_hi=g_hi, lo=a - g_lo, hi=b - g_lo)


def p1_entries(layouts, tables):
    """Entry position (global) of every rank from the exit tables: E_0 = 0, E_{r+1} = where the parse
    that enters rank r at E_r leaves it."""
    entries = []
    e = 0
    for L, X in zip(layouts, tables):
        entries.append(e)
        if e < L["b"]:
            off = e - L["a"]
            assert 0 <= off < ZONE, "entry outside the entry zone"
            e = L["b"] + X[off]
    return entries


def p1_token_split(counts):
    """From the token counts: for every rank (skip, tail) = how many of its first tokens belong to the
    left neighbour's last block, and how many tokens it needs from the right neighbour."""
    world = len(counts)
    first = [0] * world
    for r in range(1, world):
        first[r] = first[r - 1] + counts[r - 1]
    skip = [(-first[r]) % BLOCK_TOKENS if r else 0 for r in range(world)]
    for r in range(world):
        if skip[r] > counts[r]:
            raise ValueError("rank %d holds fewer tokens than one block boundary needs; use fewer ranks" % r)
    tail = [skip[r + 1] if r + 1 < world else 0 for r in range(world)]
    return skip, tail


def encode_p1_virtual(da, ctxs, data, options=None, compat=0):
    """All ranks in ONE process (one context per virtual rank, any devices): the same phases and the same
    exchanges as the distributed driver, with Python lists as the network.  Used by the GPU tests on a
    single-GPU box.  Returns the stitched stream (bytes)."""
    import torch
    options = options if options is not None else da.Compression.Default
    world = len(ctxs)
    total = len(data)
    lay = [p1_layout(total, r, world) for r in range(world)]
    bufs, shards = [], []
    for r in range(world):
        L = lay[r]
        t = torch.frombuffer(bytearray(data[L["g_lo"]:L["g_hi"]]) + bytearray(16), dtype=torch.uint8).to(
            "cuda:%d" % ctxs[r].device)
        bufs.append(t)
        shards.append(da.Shard(ctxs[r], t.data_ptr(), L["g_hi"] - L["g_lo"], L["lo"], L["hi"], L["g_lo"], total,
                               options, compat))
    tables = [s.exit_table() for s in shards]                                  # exchange 1
    entries = p1_entries(lay, tables)
    toks = [shards[r].emit(entries[r] - lay[r]["g_lo"]) for r in range(world)]  # (count, dptr)
    counts = [c for c, _ in toks]                                              # exchange 2
    skip, tail = p1_token_split(counts)
    costs = []
    for r in range(world):                                                     # exchange 3: straddling tokens
        tail_ptr = toks[r + 1][1] if tail[r] else 0
        costs.append(shards[r].blocks(skip[r], tail_ptr, tail[r]))
    allc = [c for cs in costs for c in cs]                                     # exchange 4
    plans, total_bits = da.plan_blocks(allc, compat)
    out = torch.zeros((total_bits + 7) // 8 + 16, dtype=torch.uint8)
    b0 = 0
    for r in range(world):
        nb = len(costs[r])
        mine = plans[b0:b0 + nb]
        end_bit = plans[b0 + nb][2] if b0 + nb < len(plans) else total_bits
        b0 += nb
        if not nb:
            continue
        cap = (end_bit - mine[0][2]) // 8 + 64
        dev = torch.empty(cap, dtype=torch.uint8, device=bufs[r].device)
        fb, nbytes = shards[r].pack(mine, end_bit, dev.data_ptr(), cap)
        out[fb:fb + nbytes] |= dev[:nbytes].cpu()                             # exchange 5: OR-stitch
    for s in shards:
        s.close()
    return bytes(out[: (total_bits + 7) // 8].numpy())


def encode_p1_dist(da, ctx, d_ext, layout, total, rank, world, options=None, compat=0, group=None, comm_device=None):
    """One rank of the distributed driver.  d_ext: uint8 device tensor holding bytes [g_lo, g_hi) of the
    input (+ >= 16 bytes of slack).  comm_device: where exchanged tensors live -- the GPU for the nccl
    (= RCCL) backend, "cpu" for gloo.  Returns (tensor on rank 0 | None, stream length in bytes)."""
    import torch
    options = options if options is not None else da.Compression.Default
    dev = d_ext.device
    cdev = torch.device(comm_device) if comm_device is not None else dev
    L = layout

    def gather_ints(vals):
        mine = torch.tensor(vals, dtype=torch.int64, device=cdev)
        allv = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine, group=group)
        return [t.tolist() for t in allv]

    import os
    import time
    trace = os.environ.get("MI355_P1_TRACE") == "1"
    marks = []

    def mark(name):
        if trace:
            torch.cuda.synchronize(dev)
            marks.append((name, time.perf_counter()))

    mark("start")
    sh = da.Shard(ctx, d_ext.data_ptr(), L["g_hi"] - L["g_lo"], L["lo"], L["hi"], L["g_lo"], total, options, compat,
                  torch.cuda.current_stream(dev).cuda_stream)
    mark("links+match+exit tables")
    # exchange 1: exit tables -> entry positions
    tables = gather_ints(sh.exit_table())
    lays = [p1_layout(total, r, world) for r in range(world)]
    entries = p1_entries(lays, tables)
    mark("x1 tables")
    n_tok, tok_ptr = sh.emit(entries[rank] - L["g_lo"])
    mark("emit")
    # exchange 2: token counts
    counts = [c[0] for c in gather_ints([n_tok])]
    skip, tail = p1_token_split(counts)
    # exchange 3: the head tokens of rank r+1 complete the last block of rank r
    ops = []  # (one group per rank: a send posted before the matching receive of the neighbour must not block it)
    if rank > 0 and skip[rank]:
        head = torch.empty(skip[rank], dtype=torch.int32, device=dev)
        ctypes_copy_d2d(head.data_ptr(), tok_ptr, skip[rank] * 4)
        ops.append(dist.P2POp(dist.isend, head.to(cdev), rank - 1, group))
    tail_c = None
    if tail[rank]:
        tail_c = torch.empty(tail[rank], dtype=torch.int32, device=cdev)
        ops.append(dist.P2POp(dist.irecv, tail_c, rank + 1, group))
    for q in (dist.batch_isend_irecv(ops) if ops else []):
        q.wait()
    tail_t = tail_c.to(dev) if tail_c is not None else None
    import ctypes
    import numpy as np
    mark("x2+x3 counts, straddling tokens")
    nb, carr = sh.blocks_raw(skip[rank], tail_t.data_ptr() if tail_t is not None else 0, tail[rank])
    # exchange 4: block costs (56 bytes per block) as raw bytes, padded to the largest rank; no per-block
    # Python work anywhere on this path
    mark("block costs")
    csz = ctypes.sizeof(da.BlockCost)
    nbs = [c[0] for c in gather_ints([nb])]
    mx = max(1, max(nbs))
    mine_c = torch.zeros(mx * csz, dtype=torch.uint8)
    if nb:
        mine_c[: nb * csz] = torch.from_numpy(np.frombuffer(carr, dtype=np.uint8, count=nb * csz).copy())
    mine_c = mine_c.to(cdev)
    allt = [torch.empty_like(mine_c) for _ in range(world)]
    dist.all_gather(allt, mine_c, group=group)
    flat = torch.cat([allt[r][: nbs[r] * csz] for r in range(world)]).cpu().numpy().tobytes()
    ntot = sum(nbs)
    allc = (da.BlockCost * max(1, ntot)).from_buffer_copy(flat.ljust(csz, b"\0"))
    mark("x4 costs")
    plans, total_bits = da.plan_blocks_raw(allc, ntot, compat)
    mark("plan")
    b0 = sum(nbs[:rank])
    end_bit = plans[b0 + nb].bit_start if b0 + nb < ntot else total_bits
    fb, nbytes = 0, 0
    dev_out = None
    if nb:
        cap = (end_bit - plans[b0].bit_start) // 8 + 64
        dev_out = torch.empty(cap, dtype=torch.uint8, device=dev)
        fb, nbytes = sh.pack_raw(plans, b0, end_bit, dev_out.data_ptr(), cap)
    mark("pack")
    sh.close()
    # exchange 5: byte ranges to rank 0, all in flight at once (every peer has its own xGMI link to
    # rank 0), then OR-ed in: neighbours share the seam byte
    meta = gather_ints([fb, nbytes])
    stream_len = (total_bits + 7) // 8
    if rank == 0:
        out = torch.zeros(stream_len + 16, dtype=torch.uint8, device=dev)
        if nbytes:
            out[fb:fb + nbytes] |= dev_out[:nbytes]
        tmps, ops = {}, []
        for r in range(1, world):
            f, k = meta[r]
            if k:
                tmps[r] = torch.empty(k, dtype=torch.uint8, device=cdev)
                ops.append(dist.P2POp(dist.irecv, tmps[r], r, group))
        for q in (dist.batch_isend_irecv(ops) if ops else []):
            q.wait()
        for r, t in tmps.items():
            f, k = meta[r]
            out[f:f + k] |= t.to(dev)
        mark("x5 stitch")
        if trace:
            print("P1 rank 0 phases (ms): " + ", ".join("%s %.2f" % (marks[i][0], 1e3 * (marks[i][1] - marks[i - 1][1]))
                                                      for i in range(1, len(marks))), flush=True)
        return out[:stream_len], stream_len
    if nbytes:
        for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, dev_out[:nbytes].to(cdev).contiguous(), 0, group)]):
            q.wait()
    return None, stream_len


def ctypes_copy_d2d(dst_ptr, src_ptr, nbytes):
    """device-to-device copy of raw pointers (the token array lives in the library's workspace)"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    rc = hip.hipMemcpy(ctypes.c_void_p(dst_ptr), ctypes.c_void_p(src_ptr), ctypes.c_size_t(nbytes), 3)
    if rc != 0:
        raise RuntimeError("hipMemcpy D2D failed: %d" % rc)
