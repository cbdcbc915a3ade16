"""Genome positions above 2^31 (BASELINE configs[3] / [4]: GRCh38 is 3.1 G symbols, ref_pos_t is
an unsigned 32-bit integer in the reference, common/defs.h:53).

A 3.2 G-symbol genome is built on the device; the 2^20-symbol test genome is pasted near its end,
so every window of the small-genome workloads exists again at `position + SHIFT` with
SHIFT > 3.19e9.  Each kernel must return there exactly what it returns on the small genome
(translation invariance), and the small-genome results are the oracle's."""
import numpy as np
import pytest

import common
from whisper_amd import api

pytestmark = pytest.mark.gpu

BIG_BYTES = 1_600_000_000          # 3.2 G symbols: above GRCh38's 3.1 G, below 2^32


@pytest.fixture(scope="module")
def genomes():
    import torch
    dev = torch.device("cuda", 0)
    codes, packed = common.genome()
    g = torch.Generator(device=dev).manual_seed(2024)
    big = torch.randint(0, 16, (BIG_BYTES,), dtype=torch.uint8, device=dev, generator=g)
    big = ((big >> 2) << 4) | (big & 3)             # two random symbols of A, C, G, T per byte
    at = BIG_BYTES - packed.size - 4096             # byte offset of the pasted copy (a multiple of 4)
    assert at % 4 == 0
    big[at:at + packed.size] = torch.from_numpy(packed).to(dev)
    torch.cuda.synchronize()
    small = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    large = api.Context(big, ref_size=2 * BIG_BYTES, cur_offset=0)
    yield codes, packed, small, large, 2 * at
    small.destroy()
    large.destroy()
    del big


def _shift(tasks, field, shift):
    t = tasks.copy()
    v = t[field].astype(np.int64) + shift
    assert v.min() > (1 << 31) and v.max() < (1 << 32)
    t[field] = v.astype(np.uint32)
    return t


@pytest.mark.parametrize("m,k,W,enc,last", [(150, 12, None, 0, -1), (250, 17, None, 0, -1), (100, 10, None, 1, 0),
                                            (300, 20, None, 0, -1), (150, 26, 660, 1, 0)])
def test_distance_both_modes(genomes, oracle, m, k, W, enc, last):
    codes, packed, small, large, shift = genomes
    _, _, (blob, off, lens), tasks = common.distance_workload(m, k, 5000, seed=7000 + m + k, encoding=enc, W=W)
    _, want = oracle.distance(packed, blob, off, lens, enc, tasks, last)
    big_tasks = _shift(tasks, "ref_pos", shift)
    rs, rl = small.upload_reads(blob, off, lens, enc), large.upload_reads(blob, off, lens, enc)
    np.testing.assert_array_equal(small.dynamic_programming(rs, tasks, last), want)
    np.testing.assert_array_equal(large.dynamic_programming(rl, big_tasks, last), want)
    large.set_distance_mode(api.DISTANCE_ACCEPTED_ONLY)
    got = large.dynamic_programming(rl, big_tasks, last)
    large.set_distance_mode(api.DISTANCE_EXACT)
    rej = want["edit_distance"] > k
    np.testing.assert_array_equal(got[~rej], want[~rej])
    assert (got["edit_distance"][rej] == k + 1).all() and (got["pos"][rej] == 0).all()
    rs.free()
    rl.free()


@pytest.mark.parametrize("m,k", [(150, 5), (250, 4), (150, 1)])
def test_levdiag_both_strands(genomes, oracle, m, k):
    """dir: g0 = position - cur_offset - k; other strand: g0 = ref_size - position - m + cur_offset - k
    in unsigned 32-bit arithmetic (edit_dist.cpp:112-115) with ref_size = 3.2e9."""
    codes, packed, small, large, shift = genomes
    ref_small, ref_big = codes.numel(), 2 * BIG_BYTES
    for j, (fl, fr) in enumerate(common.stage_segments(m, k)):
        cur_offset = fl
        _, _, (blob, off, lens), tasks = common.levdiag_workload(m, k, fl, fr, 2000, seed=7100 + m + 10 * k + j,
                                                                 cur_offset=cur_offset)
        p = tasks["position_in_text"].astype(np.int64) - cur_offset
        rcb = np.arange(len(tasks)) % 2 == 1
        ts, tb = tasks.copy(), tasks.copy()
        ts["position_in_text"][rcb] = (ref_small - p - m + cur_offset)[rcb]
        ts["dir_gen_flag"][rcb] = 0
        tb["position_in_text"] = (p + shift + cur_offset).astype(np.uint32)
        tb["position_in_text"][rcb] = (ref_big - (p + shift) - m + cur_offset)[rcb].astype(np.uint32)
        tb["dir_gen_flag"][rcb] = 0
        _, want = oracle.levdiag(packed, ref_small, cur_offset, blob, off, lens, ts)
        assert (want[rcb] <= k).sum() > 100 and (want[~rcb] <= k).sum() > 100
        small.set_reference_params(ref_small, cur_offset)
        large.set_reference_params(ref_big, cur_offset)
        try:
            rs, rl = small.upload_reads(blob, off, lens, 0, 1), large.upload_reads(blob, off, lens, 0, 1)
            np.testing.assert_array_equal(small.lev_diag(rs, ts), want)
            np.testing.assert_array_equal(large.lev_diag(rl, tb), want)
            rs.free()
            rl.free()
        finally:
            small.set_reference_params(ref_small, 0)
            large.set_reference_params(ref_big, 0)


def test_cf_filter_both_strands(genomes, oracle):
    codes, packed, small, large, shift = genomes
    ref_small, ref_big = codes.numel(), 2 * BIG_BYTES
    cur_offset, k = 25, 5
    _, _, (blob, off, lens), tasks = common.cf_workload(5000, seed=7200, k=k, cur_offset=cur_offset)
    _, want = oracle.cf_filter(packed, ref_small, cur_offset, blob, off, lens, tasks)
    tb = tasks.copy()
    d = tasks["dir_gen_flag"] == 1
    pos = tasks["position"].astype(np.int64)
    tb["position"] = np.where(d, pos + shift, pos + (ref_big - ref_small) - shift).astype(np.uint32)
    small.set_reference_params(ref_small, cur_offset)
    large.set_reference_params(ref_big, cur_offset)
    try:
        rs, rl = small.upload_reads(blob, off, lens, 0, 1), large.upload_reads(blob, off, lens, 0, 1)
        np.testing.assert_array_equal(small.cf_filter(rs, tasks), want)
        np.testing.assert_array_equal(large.cf_filter(rl, tb), want)
        rs.free()
        rl.free()
    finally:
        small.set_reference_params(ref_small, 0)
        large.set_reference_params(ref_big, 0)
    assert 500 < (want <= 4 * k).sum() < 4800


@pytest.mark.parametrize("m,k,W", [(150, 13, 177), (150, 26, 660), (250, 44, 700)])
def test_align(genomes, oracle, m, k, W):
    codes, packed, small, large, shift = genomes
    _, _, (blob, off, lens), tasks = common.distance_workload(m, k, 3000, seed=7300 + m + k, encoding=1, W=W)
    stride = 4 * ((2 * (m + k) + 8 + 3) // 4)
    _, want, wcig = oracle.align(packed, blob, off, lens, 1, tasks, 0, 1, cigar_stride=stride)
    rl = large.upload_reads(blob, off, lens, 1)
    got, gcig = large.align(rl, _shift(tasks, "ref_pos", shift), 0, 1, cigar_stride=stride)
    rl.free()
    for f in want.dtype.names:
        np.testing.assert_array_equal(got[f], want[f], err_msg=f)
    for i in np.nonzero(want["ok"])[0]:
        n = want["cigar_len"][i] + 1
        assert bytes(gcig[i, :n]) == bytes(wcig[i, :n]), i


@pytest.mark.parametrize("m,max_indel", [(150, 50), (300, 50), (150, 100)])
def test_indel_match_and_ext_cigar(genomes, oracle, m, max_indel):
    from oracle import bindings
    codes, packed, small, large, shift = genomes
    _, _, (blob, off, lens), tasks = common.indel_workload(2000, m, seed=7400 + m + max_indel, max_indel=max_indel,
                                                           max_mm=12 if m <= 250 else 20, stage_major=5 if m <= 250 else 8)
    prm_o, prm = bindings.default_indel_params(0.6), api.default_indel_params(0.6)
    _, want = oracle.indel_match(packed, blob, off, lens, 0, tasks, prm_o)
    rl = large.upload_reads(blob, off, lens, 0, 1)
    got = large.indel_match(rl, _shift(tasks, "ref_pos", shift), prm)
    found = want["found"] == 1
    assert found.sum() > 500
    back = got.copy()
    back["pos"][found] = (got["pos"][found].astype(np.int64) - shift).astype(np.uint32)
    np.testing.assert_array_equal(back, want)
    ct = common.indel_cigar_tasks(tasks, want)
    stride = 4 * ((2 * (m + 2 * max_indel + 20) + 8 + 3) // 4)
    _, wres, wcig = oracle.indel_ext_cigar(packed, blob, off, lens, 0, ct, cigar_stride=stride)
    gres, gcig = large.indel_ext_cigar(rl, _shift(ct, "examined_start_pos", shift), cigar_stride=stride)
    rl.free()
    np.testing.assert_array_equal(gres, wres)
    for i in range(len(wres)):
        n = wres["cigar_len"][i] + 1
        assert bytes(gcig[i, :n]) == bytes(wcig[i, :n]), i


@pytest.mark.parametrize("W,fresh", [(700, 0), (1427, 1)])
def test_softclip_match(genomes, oracle, W, fresh):
    from oracle import bindings as ob
    codes, packed, small, large, shift = genomes
    _, _, (blob, off, lens), tasks = common.softclip_workload(2000, seed=7500 + W, W=W, encoding=1,
                                                              lens_jitter=bool(fresh))
    _, want, wcig = oracle.softclip_match(packed, blob, off, lens, 1, tasks, ob.default_softclip_params(fresh_object=fresh))
    rl = large.upload_reads(blob, off, lens, 1, 3)
    got, gcig = large.softclip_match(rl, _shift(tasks, "ref_pos", shift), api.default_softclip_params(fresh_object=fresh))
    rl.free()
    assert want["found"].sum() > 1000
    np.testing.assert_array_equal(got, want)
    for i in range(len(want)):
        n = want["cigar_len"][i] + 1
        assert bytes(gcig[i, :n]) == bytes(wcig[i, :n]), i


def test_window_at_the_very_end_of_a_large_genome(genomes, oracle):
    """Windows that end in the last symbols of the 3.2 G-symbol genome: n_symbols and the clamped
    tail loads in 64-bit, positions in 32."""
    import torch
    codes, packed, small, large, shift = genomes
    dev = torch.device("cuda", 0)
    m, k = 150, 12
    W = m + 2 * k
    tail_bytes = 4096
    tail = torch.zeros(2 * tail_bytes, dtype=torch.uint8)
    g = torch.Generator().manual_seed(9)
    tail[:] = torch.randint(0, 4, (2 * tail_bytes,), generator=g, dtype=torch.uint8)
    from whisper_amd import synth
    tail_packed = synth.pack_nibbles(tail).numpy()
    # the borrowed genome tensor is the context's: write its last 4096 bytes, restore afterwards
    big = large._keep
    saved = big[-tail_bytes:].clone()
    big[-tail_bytes:] = torch.from_numpy(tail_packed).to(dev)
    torch.cuda.synchronize()
    try:
        n_t = 2 * tail_bytes
        reads_codes, tasks = [], []
        for i in range(256):
            s0 = n_t - W - 1 - (i % 40)                   # columns 1 .. W + 1 inside the tail, up to its last symbol
            p = s0 + int(torch.randint(0, 2 * k, (1,), generator=g))
            p = min(p, n_t - m)
            r = tail[p:p + m].clone()
            for _ in range(i % 7):
                x = int(torch.randint(0, m, (1,), generator=g))
                r[x] = (int(r[x]) + 1) & 3
            reads_codes.append(r)
            tasks.append((i, s0, W, k, 0, 0))
        tasks = np.array(tasks, api.DIST_TASK)
        blob, off, lens = common.mixed_reads(reads_codes, 0)
        padded = np.concatenate([tail_packed, np.full(16, 0xff, np.uint8)])   # the oracle may prefetch a few symbols
        _, want = oracle.distance(padded, blob, off, lens, 0, tasks, -1)
        assert (want["edit_distance"] <= k).sum() > 100
        big_tasks = _shift(tasks, "ref_pos", 2 * (BIG_BYTES - tail_bytes))
        rl = large.upload_reads(blob, off, lens, 0)
        np.testing.assert_array_equal(large.dynamic_programming(rl, big_tasks, -1), want)
        bad = big_tasks[:4].copy()
        bad["ref_pos"] = np.uint32(2 * BIG_BYTES - 100)   # leaves the genome: an error, not a wild read
        with pytest.raises(api.WamdError) as e:
            large.dynamic_programming(rl, bad, -1)
        assert e.value.status == 6
        rl.free()
    finally:
        big[-tail_bytes:] = saved
        torch.cuda.synchronize()
