"""Gather of per-GPU mapping records to the rank that emits SAM: the
verification path's only exchange step.  Works on any torch.distributed backend: "nccl"
(= RCCL over xGMI) on the GPUs, "gloo" in the CPU tests."""
import torch
import torch.distributed as dist

RECORD_BYTES = 12   # include/whisper_amd.h: wamd_emit_lev_records_dev


def gather_records(records, count, dst=0, group=None):
    """records: uint8 tensor [capacity, RECORD_BYTES] on this rank's device, the first
    `count` rows valid.  Returns on rank dst the list of every rank's valid rows (in rank
    order), on the other ranks None.  Two collectives: an all_gather of the counts, then a
    gather of blocks padded to the largest count."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = records.device
    cnt = torch.tensor([int(count)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    block = torch.zeros((mx, RECORD_BYTES), dtype=torch.uint8, device=dev)
    block[: counts[rank]] = records[: counts[rank]]
    out = [torch.zeros_like(block) for _ in range(world)] if rank == dst else None
    dist.gather(block, out, dst=dst, group=group)
    if rank != dst:
        return None
    return [out[r][: counts[r]] for r in range(world)]


def shard_bins(n_bins, world, rank):
    """bin b of a stage -> GPU b mod world: the bins this rank verifies."""
    return list(range(rank, n_bins, world))
