"""The C-ABI record gather over RCCL (wamd_comm_*, wamd_gather_records_dev / _all) on the one GPU a box has:
one rank gathering to itself through ncclCommInitRank / ncclCommInitAll, the all-gather of counts and the packed
layout at the root.  The multi-rank exchange itself needs more GPUs than a box has; its logic is the same loop with
ncclSend / ncclRecv in place of the local copy (tests/test_gather_gloo.py covers the N > 1 bookkeeping on the CPU)."""
import ctypes as C

import numpy as np
import pytest

import common
from whisper_amd import api

pytestmark = pytest.mark.gpu


def _records(ctx, oracle, n, seed):
    import torch
    dev = torch.device("cuda", 0)
    codes, packed, (blob, off, lens), tasks = common.distance_workload(150, 12, n, seed=seed)
    reads = ctx.upload_reads(blob, off, lens, 0)
    d_tasks = torch.from_numpy(tasks.view(np.uint8).copy()).to(dev)
    d_res = torch.zeros(2 * n, dtype=torch.int32, device=dev)
    d_pos = torch.from_numpy((tasks["ref_pos"] + 12).astype(np.uint32).view(np.int32).copy()).to(dev)
    d_dir = torch.ones(n, dtype=torch.uint8, device=dev)
    d_rec = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.dynamic_programming_dev(reads, d_tasks, n, d_res)
    ctx.emit_lev_records_static_dev(n, d_tasks, d_res, d_pos, d_dir, d_rec, d_cnt)
    ctx.synchronize()
    reads.free()
    return d_rec, d_cnt


def test_one_rank_gathers_to_itself(oracle):
    import torch
    codes, packed = common.genome()
    ctx = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    try:
        d_rec, d_cnt = _records(ctx, oracle, 20000, seed=5)
        cnt = int(d_cnt.item())
        assert cnt > 5000
        comm = api.Comm(ctx, 0, 1, api.comm_unique_id())
        d_out = torch.full((20000, 12), 0xEE, dtype=torch.uint8, device=d_rec.device)
        torch.cuda.synchronize()
        for _ in range(2):                       # a communicator serves bin after bin
            counts = comm.gather_records_dev(d_rec, d_cnt, 20000, d_out)
            ctx.synchronize()
            assert counts.tolist() == [cnt]
            assert torch.equal(d_out[:cnt], d_rec[:cnt]) and bool((d_out[cnt:] == 0xEE).all())
        with pytest.raises(api.WamdError) as e:  # a block larger than the root has room for
            comm.gather_records_dev(d_rec, d_cnt, cnt - 1, d_out)
        assert e.value.status == 8
        comm.destroy()
        # one process driving all GPUs of the node (here: the one there is)
        L = api.lib()
        ctxs = (C.c_void_p * 1)(ctx.h)
        comms = (C.c_void_p * 1)()
        assert L.wamd_comm_create_all(ctxs, C.c_int(1), comms) == 0
        recs = (C.c_void_p * 1)(d_rec.data_ptr())
        cnts = (C.c_void_p * 1)(d_cnt.data_ptr())
        d_out.fill_(0x55)
        torch.cuda.synchronize()
        counts = np.zeros(1, np.uint32)
        rc = L.wamd_gather_records_all(comms, C.c_int(1), recs, cnts, C.c_uint32(12), C.c_uint32(20000), C.c_int(0),
                                       C.c_void_p(d_out.data_ptr()), counts.ctypes.data_as(C.c_void_p), None)
        assert rc == 0, L.wamd_last_error()
        ctx.synchronize()
        assert counts.tolist() == [cnt] and torch.equal(d_out[:cnt], d_rec[:cnt])
        L.wamd_comm_destroy(C.c_void_p(comms[0]))
    finally:
        ctx.destroy()


def test_result_group_bytes_gather(oracle):
    """What rank 0 of a sharded stage actually needs: the variable-length result-group bytes of every rank's bin
    (wamd_emit_result_groups_dev leaves them and their byte count on the device) through the same gather, as records of
    one byte each."""
    import torch
    codes, packed = common.genome()
    ctx = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    try:
        t, indel, read_first, read_ids = common.commit_workload(oracle, 3000, 17)
        dev = torch.device("cuda", 0)

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)

        n, nr = len(t), len(read_ids)
        d_t, d_i, d_rf, d_ids = up(t), up(indel), up(read_first.astype(np.uint32)), up(read_ids.astype(np.uint64))
        d_c = torch.zeros(n * api.MAPPING.itemsize, dtype=torch.uint8, device=dev)
        cap = nr * 16 + n * 11 + 64
        d_bytes = torch.full((cap,), 0xEE, dtype=torch.uint8, device=dev)
        d_nb = torch.zeros(1, dtype=torch.int32, device=dev)
        d_ws = torch.zeros(ctx.result_groups_workspace_bytes(n, nr), dtype=torch.uint8, device=dev)
        d_out = torch.full((cap,), 0x55, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.commit_candidates_dev(n, d_t, d_i, d_c)
        ctx.emit_result_groups_dev(n, d_t, d_c, nr, d_rf, d_ids, api.GroupParams(5, 4, 8, 0), d_bytes, cap, d_nb, None, d_ws)
        ctx.synchronize()
        assert ctx.take_flags() == 0
        nb = int(d_nb.item())
        want = oracle_group_bytes(oracle, t, indel, read_first, read_ids)
        assert nb == len(want) and bytes(d_bytes[:nb].cpu().numpy()) == bytes(want)
        comm = api.Comm(ctx, 0, 1, api.comm_unique_id())
        counts = comm.gather_records_dev(d_bytes, d_nb, cap, d_out, record_bytes=1)
        ctx.synchronize()
        assert counts.tolist() == [nb]
        assert torch.equal(d_out[:nb], d_bytes[:nb]) and bool((d_out[nb:] == 0x55).all())
        comm.destroy()
    finally:
        ctx.destroy()


def oracle_group_bytes(oracle, t, indel, read_first, read_ids):
    from oracle import bindings as ob
    rc, committed = oracle.commit(t, indel)
    assert rc == 0
    rc, out = oracle.result_groups(t, committed, read_first, read_ids, ob.GroupParams(5, 4, 8, 0))
    assert rc == 0
    return out
