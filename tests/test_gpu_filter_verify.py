"""The CF pre-filter in front of its verifier in one call (wamd_filter_verify_distance_dev,
wamd_filter_levdiag_dev): the body of the suffix-array loops, mapping_core_sa.cpp:223-255 / 279-307.
Expected answers are composed from the two pinned oracle functions exactly as the loop composes them:
    pass = is_short_read or CF <= 4 * max_mismatches;  cur_match = pass ? verifier : max_mismatches + 1.
Bit-exact bar; both distance modes; the survivors are packed on the device, so the shapes include batches
where nothing passes, everything passes, and lengths that are not multiples of a block."""
import numpy as np
import pytest
import torch

import common
from whisper_amd import api, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    codes, packed = common.genome()
    c = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    yield c
    c.destroy()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to("cuda:0")


def workspace(ctx, n):
    wsb = ctx.filter_verify_workspace_bytes(n)
    d_ws = torch.zeros(wsb + 256, dtype=torch.uint8, device="cuda:0")
    pad = (-d_ws.data_ptr()) % 256
    return d_ws[pad:], wsb


def candidates(n, seed, k, min_len, cur_offset, short_every=7):
    """cf_workload's candidates with the verifier tasks of the same loci next to them."""
    codes, packed, (blob, off, lens), cf = common.cf_workload(n, seed=seed, k=k, cur_offset=cur_offset, min_len=min_len)
    G = codes.numel()
    direct = cf["dir_gen_flag"] == 1
    pos = cf["position"].astype(np.int64) - cur_offset
    p = np.where(direct, pos, G - pos - min_len)                   # genome symbol under the window's first symbol
    start = np.where(direct, p, p - (lens.astype(np.int64) - min_len))   # ... under the oriented read's first symbol
    dist = np.zeros(n, api.DIST_TASK)
    dist["read_idx"] = cf["read_idx"]
    dist["ref_pos"] = (start - k).astype(np.uint32)
    dist["max_distance_in_ref"] = lens + 2 * k
    dist["max_ed"] = k
    dist["flags"] = cf["flags"]
    lev = np.zeros(n, api.LEVDIAG_TASK)
    lev["read_idx"] = cf["read_idx"]
    lev["position_in_text"] = (start + cur_offset).astype(np.uint32)
    lev["fixed_left"] = cf["fixed_left"]
    lev["fixed_right"] = cf["fixed_right"]
    lev["max_mismatches"] = k
    lev["dir_gen_flag"] = 1
    short = np.arange(n) % short_every == 3 if short_every else np.zeros(n, bool)
    return codes, packed, (blob, off, lens), cf, dist, lev, short


def cf_pass(oracle, packed, G, cur_offset, reads, cf, short, k):
    rc, val = oracle.cf_filter(packed, G, cur_offset, *reads, cf)
    assert rc == 0
    return short | (val <= 4 * k)


@pytest.mark.parametrize("k,cur_offset,min_len,n", [(5, 0, 150, 6000), (12, 17, 150, 5003), (7, 0, 250, 3000),
                                                    (4, 9, 100, 4097), (1, 0, 150, 2000),
                                                    (6, 0, 300, 2000),    # M64 rule, 64-thread blocks of the exact kernel
                                                    (3, 5, 40, 2500)])    # two dwords per vector: no banded kernels
def test_filter_then_distance(ctx, oracle, k, cur_offset, min_len, n):
    codes, packed, reads_h, cf, dist, lev, short = candidates(n, 400 + k, k, min_len, cur_offset)
    G = codes.numel()
    passed = cf_pass(oracle, packed, G, cur_offset, reads_h, cf, short, k)
    rc, want = oracle.distance(packed, *reads_h, 0, dist, -1)
    assert rc == 0
    acc = passed & (want["edit_distance"] <= k)
    assert acc.sum() > n // 8 and (passed & ~acc).sum() > 20 and (~passed).sum() > n // 8
    assert (~passed & (want["edit_distance"] <= k)).sum() >= 0   # the filter may turn away what the verifier would accept
    cf_dev = cf.copy()
    cf_dev["pattern_len"][short] = 0
    ctx.set_reference_params(G, cur_offset)
    try:
        reads = ctx.upload_reads(*reads_h, 0, orientations=3)
        d_cf, d_t = dev(cf_dev), dev(dist)
        d_ws, wsb = workspace(ctx, n)
        for mode in (api.DISTANCE_EXACT, api.DISTANCE_ACCEPTED_ONLY):
            ctx.set_distance_mode(mode)
            d_r = torch.full((n * 2,), 0x55, dtype=torch.int32, device="cuda:0")
            ctx.filter_verify_distance_dev(reads, d_cf, d_t, n, d_r, d_ws, wsb, last_symbol=-1)
            ctx.synchronize()
            assert ctx.take_flags() == 0
            got = d_r.cpu().numpy().view(np.uint32).reshape(-1, 2)
            exp_ed = np.where(passed, want["edit_distance"], k + 1)
            exp_pos = np.where(passed, want["pos"], 0)
            if mode == api.DISTANCE_ACCEPTED_ONLY:
                rej = exp_ed > k
                exp_ed = np.where(rej, k + 1, exp_ed)
                exp_pos = np.where(rej, 0, exp_pos)
            np.testing.assert_array_equal(got[:, 0], exp_ed)
            np.testing.assert_array_equal(got[:, 1], exp_pos)
        reads.free()
    finally:
        ctx.set_distance_mode(api.DISTANCE_EXACT)
        ctx.set_reference_params(G, 0)


@pytest.mark.parametrize("k,cur_offset,min_len,n", [(5, 0, 150, 6000), (2, 17, 150, 3001), (4, 0, 250, 3000), (6, 9, 100, 4000),
                                                    (4, 0, 300, 2000),    # above 256 symbols: the run-extension kernel alone
                                                    (5, 0, 600, 1500)])
def test_filter_then_levdiag(ctx, oracle, k, cur_offset, min_len, n):
    codes, packed, reads_h, cf, dist, lev, short = candidates(n, 500 + k, k, min_len, cur_offset)
    G = codes.numel()
    passed = cf_pass(oracle, packed, G, cur_offset, reads_h, cf, short, k)
    rc, want = oracle.levdiag(packed, G, cur_offset, *reads_h, lev)
    assert rc == 0
    assert (passed & (want <= k)).sum() > n // 16 and (passed & (want > k)).sum() > 20 and (~passed).sum() > n // 8
    cf_dev = cf.copy()
    cf_dev["pattern_len"][short] = 0
    ctx.set_reference_params(G, cur_offset)
    try:
        reads = ctx.upload_reads(*reads_h, 0, orientations=3)
        d_ws, wsb = workspace(ctx, n)
        d_r = torch.full((n,), 0x55, dtype=torch.int32, device="cuda:0")
        ctx.filter_levdiag_dev(reads, dev(cf_dev), dev(lev), n, d_r, d_ws, wsb)
        ctx.synchronize()
        assert ctx.take_flags() == 0
        reads.free()
    finally:
        ctx.set_reference_params(G, 0)
    np.testing.assert_array_equal(d_r.cpu().numpy().view(np.uint32), np.where(passed, want, k + 1))


def test_filter_extremes_and_errors(ctx, oracle):
    """Nothing passes / everything passes (short reads) / one candidate; a malformed CF task raises its flag
    and leaves the error value in its slot without disturbing its neighbours; a small workspace is refused."""
    k, n = 5, 1000
    codes, packed, reads_h, cf, dist, lev, short = candidates(n, 77, k, 150, 0, short_every=0)
    G = codes.numel()
    rc, want = oracle.distance(packed, *reads_h, 0, dist, -1)
    passed = cf_pass(oracle, packed, G, 0, reads_h, cf, short, k)
    reads = ctx.upload_reads(*reads_h, 0, orientations=3)
    d_ws, wsb = workspace(ctx, n)

    def run(cf_t, dist_t, count):
        d_r = torch.full((count * 2,), 0x55, dtype=torch.int32, device="cuda:0")
        ctx.filter_verify_distance_dev(reads, dev(cf_t), dev(dist_t), count, d_r, d_ws, wsb)
        ctx.synchronize()
        return d_r.cpu().numpy().view(np.uint32).reshape(-1, 2)

    none = ~passed
    got = run(cf[none], dist[none], int(none.sum()))
    assert ctx.take_flags() == 0 and (got[:, 0] == k + 1).all() and (got[:, 1] == 0).all()
    all_short = cf.copy()
    all_short["pattern_len"] = 0
    got = run(all_short, dist, n)
    assert ctx.take_flags() == 0
    np.testing.assert_array_equal(got[:, 0], want["edit_distance"])
    np.testing.assert_array_equal(got[:, 1], want["pos"])
    one = int(np.flatnonzero(passed)[0])
    got = run(cf[one:one + 1], dist[one:one + 1], 1)
    assert tuple(got[0]) == (want["edit_distance"][one], want["pos"][one])
    bad = cf.copy()
    bad["fixed_left"][10] = 1
    got = run(bad, dist, n)
    assert ctx.take_flags() != 0 and got[10, 0] == 0xFFFFFFFF
    keep = np.arange(n) != 10
    np.testing.assert_array_equal(got[keep, 0], np.where(passed, want["edit_distance"], k + 1)[keep])
    with pytest.raises(api.WamdError):
        d_r = torch.zeros(n * 2, dtype=torch.int32, device="cuda:0")
        ctx.filter_verify_distance_dev(reads, dev(cf), dev(dist), n, d_r, d_ws, 1024)
    reads.free()


def test_filter_verify_equals_the_two_calls_at_scale():
    """Size-independent property at 2^20 candidates (8 per read, one of them the read's locus): the fused call
    gives what wamd_cf_filter_dev followed by wamd_verify_distance_dev give, in both distance modes."""
    devc = torch.device("cuda", 0)
    na, c_per, k, m = 1 << 17, 8, 12, 150
    n = na * c_per
    codes = synth.make_genome(1 << 24, seed=78, device=devc)
    packed = synth.pack_nibbles(codes)
    G = codes.numel()
    c = api.Context(packed, ref_size=G, cur_offset=0, device=0)
    try:
        rd = synth.make_reads(codes, na, m, k, seed=3131, chunk=1 << 15)
        blob, off, lens = synth.pack_reads(rd["codes"], 0)
        reads = c.upload_reads(blob.cpu().numpy(), off.cpu().numpy().astype(np.uint64),
                               lens.cpu().numpy().astype(np.uint32), api.READS_NIBBLE, orientations=3)
        rng = np.random.default_rng(5)
        p0 = rd["p"].cpu().numpy().astype(np.int64)
        is_rc = rd["is_rc"].cpu().numpy()
        loci = rng.integers(1 << 16, G - (1 << 16), n)
        loci[0::c_per] = p0
        dist = np.zeros(n, api.DIST_TASK)
        dist["read_idx"] = np.repeat(np.arange(na, dtype=np.uint32), c_per)
        dist["ref_pos"] = (loci - k).astype(np.uint32)
        dist["max_distance_in_ref"] = m + 2 * k
        dist["max_ed"] = k
        dist["flags"] = np.repeat(is_rc.astype(np.uint8), c_per)
        cf = np.zeros(n, api.CF_TASK)
        cf["read_idx"] = dist["read_idx"]
        cf["pattern_len"] = m
        rcs = np.repeat(is_rc, c_per)
        cf["position"] = np.where(rcs, G - loci - m, loci).astype(np.uint32)
        cf["fixed_left"] = np.where(rcs, m - 75, 50)
        cf["fixed_right"] = np.where(rcs, m - 50, 75)
        cf["dir_gen_flag"] = ~rcs
        cf["flags"] = rcs
        d_cf, d_t = dev(cf), dev(dist)
        d_v = torch.zeros(n, dtype=torch.int32, device=devc)
        c.cf_filter_dev(reads, d_cf, n, d_v)
        c.synchronize()
        passed = d_v.cpu().numpy().view(np.uint32) <= 4 * k
        # (4k = 48 is a loose bound: about half of the unrelated loci pass it too)
        assert passed[0::c_per].mean() > 0.6 and 0.2 < passed.mean() < 0.8
        d_ws, wsb = workspace(c, n)
        for mode in (api.DISTANCE_EXACT, api.DISTANCE_ACCEPTED_ONLY):
            c.set_distance_mode(mode)
            d_a = torch.zeros(n * 2, dtype=torch.int32, device=devc)
            d_b = torch.zeros(n * 2, dtype=torch.int32, device=devc)
            c.dynamic_programming_dev(reads, d_t, n, d_a, last_symbol=-1)
            c.filter_verify_distance_dev(reads, d_cf, d_t, n, d_b, d_ws, wsb, last_symbol=-1)
            c.synchronize()
            assert c.take_flags() == 0
            a = d_a.cpu().numpy().view(np.uint32).reshape(-1, 2)
            b = d_b.cpu().numpy().view(np.uint32).reshape(-1, 2)
            np.testing.assert_array_equal(b[passed], a[passed])
            assert (b[~passed, 0] == k + 1).all() and (b[~passed, 1] == 0).all()
        reads.free()
    finally:
        c.destroy()
