"""Sharding one input over the GPUs of a node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for the tests).

Each rank owns a contiguous byte range of the input and encodes it as one chunk of the final
stream: every rank but the last ends its chunk with the reference's sync-flush form (all blocks
non-final + empty stored block, byte aligned; include/mi355_deflate.h MI355_FLUSH_SYNC), the last
rank finishes normally.  The only exchange on the data path is the final stitch: an all-gather of
the chunk sizes (8 bytes per rank) and point-to-point sends of the compressed chunks into rank 0's
output buffer at their byte offsets.  The result equals
    concat_i  fresh_reference_encoder(chunk_i).write_all().flush()   (finish() for the last)
i.e. it is chunk-exact ("P2" in SURVEY.md section 0), not identical to the reference run on the
whole input: matches do not cross rank boundaries.
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
