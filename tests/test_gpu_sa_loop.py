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


def _compact(z):
    """The fixture's candidates as the compact form holds them: counts per read and strand, the SA entries."""
    nr = len(z["read_ids"])
    rf = z["read_first"].astype(np.int64)
    direct = z["cf"]["dir_gen_flag"].astype(np.int64)
    counts = np.zeros((nr, 2), np.uint16)
    for r in range(nr):
        d = int(direct[rf[r]:rf[r + 1]].sum())
        counts[r] = (d, rf[r + 1] - rf[r] - d)
    return counts, z["cf"]["position"].astype(np.uint32)


@pytest.mark.parametrize("filtered", [True, False])
@pytest.mark.parametrize("name", gu.cases("saloop"))
def test_compact_form_equals_the_task_form(name, filtered):
    """wamd_verify_sa_candidates(_dev): counts + 4-byte suffix-array entries expanded to tasks on the device must give
    what the task form gives for the tasks the fixture holds (which the test above pins to the reference's loop):
    errors clamped at k + 1, end columns of the accepted candidates; the host-pointer pipeline the same again."""
    z = gu.load(name)
    k, seg, mrl = int(z["k"]), (int(z["seg"][0]), int(z["seg"][1])), int(z["min_read_len"])
    n, nr = len(z["commit"]), len(z["read_ids"])
    lev = k < 7
    counts, sa = _compact(z)
    cf = z["cf"].copy()
    if not filtered:
        cf["pattern_len"] = 0          # the task form's "short read": no filter
    ctx = api.Context(z["packed"], ref_size=int(z["ref_size"]), cur_offset=seg[0])
    try:
        ctx.set_distance_mode(api.DISTANCE_EXACT)
        reads = ctx.upload_reads(z["blob"], z["off"], z["lens"], api.READS_NIBBLE, orientations=3)
        wsb = ctx.filter_verify_workspace_bytes(n)
        d_ws = torch.zeros(wsb + 256, dtype=torch.uint8, device="cuda:0")
        d_ws = d_ws[(-d_ws.data_ptr()) % 256:]
        d_cf, d_lev, d_dist = dev(cf), dev(z["lev"]), dev(z["dist"])
        d_ed = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        d_res = torch.zeros(n * 2, dtype=torch.int32, device="cuda:0")
        b = api.sa_batch(nr, k, seg, mrl if filtered else 0, api.SA_VERIFIER_LEVDIAG if lev else api.SA_VERIFIER_DISTANCE,
                         reverse_read_offset=nr)
        wsb2 = ctx.sa_candidates_workspace_bytes(n, nr)
        d_ws2 = torch.zeros(wsb2 + 256, dtype=torch.uint8, device="cuda:0")
        d_ws2 = d_ws2[(-d_ws2.data_ptr()) % 256:]
        d_counts, d_sa = dev(counts), dev(sa)
        d_err = torch.full((n + 3,), 0xEE, dtype=torch.uint8, device="cuda:0")
        d_end = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        ts = torch.cuda.Stream()
        st = ts.cuda_stream
        with torch.cuda.stream(ts):
            if lev:
                ctx.filter_levdiag_dev(reads, d_cf, d_lev, n, d_ed, d_ws, wsb, stream=st)
            else:
                ctx.filter_verify_distance_dev(reads, d_cf, d_dist, n, d_res, d_ws, wsb, last_symbol=-1, stream=st)
            ctx.verify_sa_candidates_dev(reads, b, d_counts, d_sa, n, d_err, None if lev else d_end, d_ws2, wsb2, stream=st)
        torch.cuda.synchronize()
        assert ctx.take_flags(st) == 0
        if lev:
            want_e, want_p = d_ed.cpu().numpy().view(np.uint32), None
        else:
            r = d_res.cpu().numpy().view(np.uint32).reshape(n, 2)
            want_e, want_p = r[:, 0], np.where(r[:, 0] <= k, r[:, 1], 0)
        want_e = np.minimum(want_e, k + 1).astype(np.uint8)
        got = d_err.cpu().numpy()
        np.testing.assert_array_equal(got[:n], want_e)
        assert (got[n:] == 0xEE).all() and (want_e <= k).sum() > n // 10
        if want_p is not None:
            np.testing.assert_array_equal(d_end.cpu().numpy().view(np.uint32), want_p)
        # host buffers (page-locked and pageable) through the chunked pipeline
        pc, ps, pe = api.PinnedBuffer(2 * nr, np.uint16), api.PinnedBuffer(n, np.uint32), api.PinnedBuffer(n, np.uint8)
        pc.array[:] = counts.reshape(-1)
        ps.array[:] = sa
        pe.array[:] = 0xEE
        ctx.verify_sa_candidates(reads, b, pc.array, ps.array, errors=pe.array)
        np.testing.assert_array_equal(pe.array, want_e)
        for buf in (pc, ps, pe):
            buf.free()
        out = ctx.verify_sa_candidates(reads, b, counts, sa, want_positions=not lev)
        if lev:
            np.testing.assert_array_equal(out, want_e)
        else:
            np.testing.assert_array_equal(out[0], want_e)
            np.testing.assert_array_equal(out[1], want_p)
        # mapping_mode second: every read verifies with its own k = min(stage, best_error + 1) (mapping_core.cpp:955-956)
        rng = np.random.default_rng(3)
        read_k = rng.integers(max(1, k - 4) if lev else max(7, k - 5), k + 1, nr).astype(np.uint8)
        per_cand = np.repeat(read_k, counts.sum(axis=1).astype(np.int64))
        lev_t, dist_t = z["lev"].copy(), z["dist"].copy()
        lev_t["max_mismatches"] = per_cand
        dist_t["max_ed"] = per_cand
        # the loops take the window from the same max_mismatches (mapping_core_sa.cpp:237: sa - start_pos - k, len + 2k)
        dist_t["ref_pos"] = dist_t["ref_pos"] + (k - per_cand)
        dist_t["max_distance_in_ref"] = dist_t["max_distance_in_ref"] - 2 * (k - per_cand.astype(np.uint32))
        d_lev2, d_dist2 = dev(lev_t), dev(dist_t)
        torch.cuda.synchronize()
        with torch.cuda.stream(ts):
            if lev:
                ctx.filter_levdiag_dev(reads, d_cf, d_lev2, n, d_ed, d_ws, wsb, stream=st)
            else:
                ctx.filter_verify_distance_dev(reads, d_cf, d_dist2, n, d_res, d_ws, wsb, last_symbol=-1, stream=st)
        torch.cuda.synchronize()
        assert ctx.take_flags(st) == 0
        e2 = d_ed.cpu().numpy().view(np.uint32) if lev else d_res.cpu().numpy().view(np.uint32).reshape(n, 2)[:, 0]
        want2 = np.where(e2 <= per_cand, e2, per_cand.astype(np.uint32) + 1).astype(np.uint8)
        got2 = ctx.verify_sa_candidates(reads, b, counts, sa, read_k=read_k)
        np.testing.assert_array_equal(got2, want2)
        assert (want2 != want_e).any() or (read_k == k).all()     # (the per-read k changed something, where it could)
        # counts that do not add up are an error, on either path
        bad = counts.copy()
        bad[0, 0] += 1
        with pytest.raises(api.WamdError):
            ctx.verify_sa_candidates(reads, b, bad, sa)
        reads.free()
    finally:
        ctx.destroy()
