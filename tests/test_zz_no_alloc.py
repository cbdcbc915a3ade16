"""wamd_emit_lev_records_static_dev (the entry point that allocates nothing: scratch in the
library's static device array) against wamd_emit_lev_records_dev, byte for byte, on two calls
in a row - the second with a smaller batch, so stale counts of the first call sit behind the
words the second one uses."""
import numpy as np
import pytest

import common
from whisper_amd import api

pytestmark = pytest.mark.gpu


def _emit(ctx, static, n, d_tasks, d_res, d_pos, d_dir):
    import torch
    dev = d_tasks.device
    d_rec = torch.full((max(n, 1), 12), 0xEE, dtype=torch.uint8, device=dev)
    d_cnt = torch.full((1,), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    if static:
        ctx.emit_lev_records_static_dev(n, d_tasks, d_res, d_pos, d_dir, d_rec, d_cnt)
    else:
        d_scr = torch.zeros(ctx.records_scratch_bytes(n), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.emit_lev_records_dev(n, d_tasks, d_res, d_pos, d_dir, d_rec, d_cnt, d_scr)
    ctx.synchronize()
    assert ctx.take_flags() == 0
    cnt = int(d_cnt.item())
    return cnt, d_rec.cpu().numpy()[:cnt]


def test_static_records_equal_the_workspace_form():
    import torch
    codes, packed = common.genome()
    ctx = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    try:
        dev = torch.device("cuda", 0)
        for n, seed in ((70001, 91), (5003, 92), (0, 93)):   # second call smaller, third empty
            _, _, (blob, off, lens), tasks = common.distance_workload(150, 12, max(n, 1), seed=seed)
            tasks = tasks[:n]
            reads = ctx.upload_reads(blob, off, lens, 0)
            d_tasks = torch.from_numpy(tasks.view(np.uint8).copy()).to(dev)
            d_res = torch.zeros(2 * max(n, 1), dtype=torch.int32, device=dev)
            pos = (tasks["ref_pos"] + 12).astype(np.uint32)
            direction = (np.arange(n) % 3 != 0).astype(np.uint8)
            d_pos = torch.from_numpy(pos.view(np.int32).copy()).to(dev)
            d_dir = torch.from_numpy(direction).to(dev)
            torch.cuda.synchronize()
            if n:
                ctx.dynamic_programming_dev(reads, d_tasks, n, d_res)
                ctx.synchronize()
            want_cnt, want = _emit(ctx, False, n, d_tasks, d_res, d_pos, d_dir)
            got_cnt, got = _emit(ctx, True, n, d_tasks, d_res, d_pos, d_dir)
            reads.free()
            assert got_cnt == want_cnt
            if n:
                assert want_cnt > n // 4
            np.testing.assert_array_equal(got, want)
        # beyond the static scratch: a clear error, nothing launched
        dummy = torch.zeros(64, dtype=torch.int32, device=dev)
        with pytest.raises(api.WamdError) as e:
            ctx.emit_lev_records_static_dev((1 << 24) + 1, dummy, dummy, dummy, dummy, dummy, dummy)
        assert e.value.status == 8
    finally:
        ctx.destroy()
