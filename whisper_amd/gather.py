"""Gather of per-GPU mapping records to the rank that emits SAM: the
verification path's only exchange step.  Works on any torch.distributed backend: "nccl"
(= RCCL over xGMI) on the GPUs, "gloo" in the CPU tests."""
import torch
import torch.distributed as dist

RECORD_BYTES = 12   # include/whisper_amd.h: wamd_emit_lev_records_dev


class PendingGather:
    """A gather in flight (gather_records(..., async_op=True)): result() waits for it and
    returns what gather_records would have; the buffers live as long as this object."""

    def __init__(self, work, out, counts, rank, dst, block):
        self.work, self.out, self.counts, self.rank, self.dst, self.block = work, out, counts, rank, dst, block

    def result(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.rank != self.dst:
            return None
        return [self.out[r][: self.counts[r]] for r in range(len(self.counts))]


_OUT_CACHE = {}


def gather_records(records, count, dst=0, group=None, async_op=False, copy=True):
    """records: uint8 tensor [capacity, RECORD_BYTES] on this rank's device, the first
    `count` rows valid (count: an int, or a one-element integer tensor on that device - then
    no host round trip precedes the first collective).  Returns on rank dst the list of every
    rank's valid rows (in rank order), on the other ranks None.  Two collectives: an all_gather
    of the counts, then a gather of blocks padded to the largest count.  With async_op the
    gather itself is left in flight (a PendingGather): the caller verifies its next bin
    meanwhile.  copy=False sends records[:largest count] itself instead of a padded copy (every
    rank's capacity must cover the largest count, and the caller leaves `records` alone until
    result()); the receive buffers of rank dst are then reused from call to call, so a result
    is only valid until the next call."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = records.device
    if torch.is_tensor(count):
        cnt = count.reshape(1).to(device=dev, dtype=torch.int64)
    else:
        cnt = torch.tensor([int(count)], dtype=torch.int64, device=dev)
    counts = [torch.empty_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = torch.cat(counts).tolist()   # one read-back
    mx = max(max(counts), 1)
    if copy:
        block = torch.zeros((mx, RECORD_BYTES), dtype=torch.uint8, device=dev)
        block[: counts[rank]] = records[: counts[rank]]
        out = [torch.zeros_like(block) for _ in range(world)] if rank == dst else None
    else:
        if records.shape[0] < mx:
            raise ValueError("gather_records(copy=False): capacity %d below the largest count %d" % (records.shape[0], mx))
        block = records[:mx]
        out = None
        if rank == dst:
            key = (dev, world)
            buf = _OUT_CACHE.get(key)
            if buf is None or buf.shape[1] < mx:
                buf = torch.empty((world, max(mx, records.shape[0]), RECORD_BYTES), dtype=torch.uint8, device=dev)
                _OUT_CACHE[key] = buf
            out = [buf[r, :mx] for r in range(world)]
    work = dist.gather(block, out, dst=dst, group=group, async_op=async_op)
    pending = PendingGather(work if async_op else None, out, counts, rank, dst, block)
    return pending if async_op else pending.result()


def shard_bins(n_bins, world, rank):
    """bin b of a stage -> GPU b mod world: the bins this rank verifies."""
    return list(range(rank, n_bins, world))
