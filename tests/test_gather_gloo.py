"""The N > 1 path on CPU: two processes (gloo) shard bins and gather variable-length
mapping-record blocks to rank 0, as bench.py does over RCCL."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from whisper_amd import gather


def _records(rank, n):
    g = np.random.default_rng(100 + rank)
    r = np.zeros((n + 5, gather.RECORD_BYTES), np.uint8)
    r[:n] = g.integers(0, 256, (n, gather.RECORD_BYTES), dtype=np.uint8)
    return r


def _worker(rank, world, port, counts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rec = torch.from_numpy(_records(rank, counts[rank]))
        out = gather.gather_records(rec, counts[rank], dst=0)
        if rank == 0:
            q.put([o.numpy().copy() for o in out])
        else:
            assert out is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker_pipelined(rank, world, port, rounds, q):
    """bench.py's pattern: the gather of bin i is in flight while bin i + 1 is produced."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pending, outs = None, []
        for i, counts in enumerate(rounds):
            rec = torch.from_numpy(_records(rank + 10 * i, counts[rank]))
            nxt = gather.gather_records(rec, counts[rank], dst=0, async_op=True)
            rec.zero_()                      # the caller may overwrite its buffer at once
            if pending is not None:
                outs.append(pending.result())
            pending = nxt
        outs.append(pending.result())
        # the copy-free form bench.py uses: count as a tensor, the record buffer itself is sent
        # (capacity 16 on every rank), results valid until the next call
        for i, counts in enumerate(rounds):
            rec = torch.zeros((16, gather.RECORD_BYTES), dtype=torch.uint8)
            rec[: counts[rank]] = torch.from_numpy(_records(rank + 10 * i, counts[rank]))[: counts[rank]]
            out = gather.gather_records(rec, torch.tensor([counts[rank]], dtype=torch.int32), dst=0,
                                        async_op=True, copy=False).result()
            if rank == 0:
                for r in range(world):
                    np.testing.assert_array_equal(out[r].numpy(), _records(r + 10 * i, counts[r])[: counts[r]])
            else:
                assert out is None
        if rank == 0:
            q.put([[o.numpy().copy() for o in out] for out in outs])
        else:
            assert all(o is None for o in outs)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gather_two_ranks():
    for counts in ([7, 3], [0, 11], [4, 0]):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, counts, q)) for r in range(2)]
        for p in procs:
            p.start()
        got = q.get(timeout=120)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        for r in range(2):
            np.testing.assert_array_equal(got[r], _records(r, counts[r])[: counts[r]])


def test_gather_pipelined_two_ranks():
    rounds = [[7, 3], [0, 11], [5, 5], [4, 0]]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipelined, args=(r, 2, port, rounds, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for i, counts in enumerate(rounds):
        for r in range(2):
            np.testing.assert_array_equal(got[i][r], _records(r + 10 * i, counts[r])[: counts[r]])


def test_shard_bins_partition():
    for world in (1, 2, 4, 8):
        seen = sorted(b for r in range(world) for b in gather.shard_bins(384, world, r))
        assert seen == list(range(384))


def _run_bench(args, env_extra=None):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, cwd=root,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)


def test_bench_starts_its_own_ranks_and_refuses_a_cpu_run():
    """`python bench.py --gpus 2` with no launcher starts two ranks as a child torch.distributed.run; on a machine
    without a GPU each of them stops with the "needs a GPU" message (no CPU fallback), nothing reaches stdout and the
    exit status is not 0.  (Skipped where a GPU is present: there the same command is a real two-rank run.)"""
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    p = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0
    assert p.stdout.strip() == b""
    assert p.stderr.count(b"bench.py needs a GPU") >= 2


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    p = _run_bench(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0
    assert p.stdout.strip() == b""
    assert b"--gpus 2 but the launcher started 1 rank" in p.stderr
