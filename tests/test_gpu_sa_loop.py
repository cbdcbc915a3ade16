"""The device leg of the mapping loop against the reference's loop itself.  Fixtures tests/golden/saloop_*: what
CMappingCore::search_in_unknown_range_dir_and_rc / search_in_known_range_dir_and_rc (mapping_core_sa.cpp:193-513) pushed
to dir / rc_match_positions for sorted reads over a real suffix array, and the bytes CMappingsHeapGatherer +
CMappingResultsCollector made of those pushes.  Here the same suffix-array entries go through the _dev entry points,
nothing leaving the device in between:
    wamd_filter_levdiag_dev | wamd_filter_verify_distance_dev   the loop body's CF test + verifier (:223-255, 279-307)
    wamd_indel_match_dev                                        CIndelMatching::Match in sensitive stages (:258-272)
    wamd_commit_candidates_dev                                  construct<lev>, better, the penalty test (:241-247, 334-344)
    wamd_emit_result_groups_dev                                 mapping_core.cpp:976-997, results.cpp
and must yield the same mappings in the same order and the same bytes."""
import numpy as np
import pytest
import torch

import golden_util as gu
from whisper_amd import api

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to("cuda:0")


@pytest.mark.parametrize("mode", [api.DISTANCE_EXACT, api.DISTANCE_ACCEPTED_ONLY])
@pytest.mark.parametrize("name", gu.cases("saloop"))
def test_device_loop_equals_the_reference_loop(name, mode):
    z = gu.load(name)
    k, fl = int(z["k"]), int(z["seg"][0])
    n = len(z["commit"])
    nr = len(z["read_ids"])
    ctx = api.Context(z["packed"], ref_size=int(z["ref_size"]), cur_offset=fl)     # CUR_OFFSET = the anchor's start
    try:
        ctx.set_distance_mode(mode)
        reads = ctx.upload_reads(z["blob"], z["off"], z["lens"], api.READS_NIBBLE, orientations=3)
        wsb = ctx.filter_verify_workspace_bytes(n)
        d_ws = torch.zeros(wsb + 256, dtype=torch.uint8, device="cuda:0")
        d_ws = d_ws[(-d_ws.data_ptr()) % 256:]
        # every device buffer exists (and torch's own stream is idle) before the first call on the context's stream,
        # and stays referenced until the context has been synchronised
        d_commit = dev(z["commit"]).view(torch.int32).reshape(n, 4)
        d_cf, d_lev, d_dist, d_it = dev(z["cf"]), dev(z["lev"]), dev(z["dist"]), dev(z["indel"])
        d_ed = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        d_res = torch.zeros(n * 2, dtype=torch.int32, device="cuda:0")
        d_indel = torch.zeros(n * api.MAPPING.itemsize, dtype=torch.uint8, device="cuda:0")
        d_c = torch.zeros(n * api.MAPPING.itemsize, dtype=torch.uint8, device="cuda:0")
        cap = nr * 16 + n * 11 + 64
        d_out = torch.full((cap,), 0xEE, dtype=torch.uint8, device="cuda:0")
        d_nb = torch.zeros(1, dtype=torch.int32, device="cuda:0")
        d_wsg = torch.zeros(ctx.result_groups_workspace_bytes(n, nr), dtype=torch.uint8, device="cuda:0")
        d_rf, d_ids = dev(z["read_first"].astype(np.uint32)), dev(z["read_ids"].astype(np.uint64))
        sensitive = int(z["sensitive"])
        if sensitive:
            d_commit[:, 3] |= 1 << 8          # has_indel (byte 13 of the task)
        torch.cuda.synchronize()
        ts = torch.cuda.Stream()                          # one stream for the library's kernels and torch's copies
        st = ts.cuda_stream
        with torch.cuda.stream(ts):
            if k < 7:
                ctx.filter_levdiag_dev(reads, d_cf, d_lev, n, d_ed, d_ws, wsb, stream=st)
                d_commit[:, 1] = d_ed
            else:
                ctx.filter_verify_distance_dev(reads, d_cf, d_dist, n, d_res, d_ws, wsb, last_symbol=-1, stream=st)
                d_commit[:, 1] = d_res.reshape(n, 2)[:, 0]
            if sensitive:
                ctx.indel_match_dev(reads, d_it, n, d_indel, api.default_indel_params(float(z["min_clipped_factor"])),
                                    stream=st)
            ctx.commit_candidates_dev(n, d_commit, d_indel if sensitive else None, d_c, stream=st)
            ctx.emit_result_groups_dev(n, d_commit, d_c, nr, d_rf, d_ids, api.GroupParams(5, 4, 1024, 0), d_out, cap, d_nb,
                                       None, d_wsg, stream=st)
        torch.cuda.synchronize()
        assert ctx.take_flags(st) == 0
        reads.free()
        committed = d_c.cpu().numpy().view(api.MAPPING)
        commit = d_commit.cpu().numpy().reshape(-1).view(api.COMMIT_TASK)
        kept = committed["found"] == 1
        assert kept.sum() == len(z["pushed"])
        for f in ("type", "pos", "no_mismatches", "matching_len1", "matching_len2", "indel_len1", "indel_len2",
                  "clipping_len", "penalty"):
            np.testing.assert_array_equal(committed[f][kept], z["pushed"][f], err_msg=f)
        np.testing.assert_array_equal(commit["direct"][kept], z["pushed_direct"])
        got = d_out.cpu().numpy()[:int(d_nb.item())]
        assert bytes(got) == bytes(z["group_bytes"])
    finally:
        ctx.destroy()
