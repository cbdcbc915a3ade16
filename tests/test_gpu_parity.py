"""HIP path (through the C-ABI) vs the oracle on seeded tasks.  Bit-exact bar for edit
distances, positions, CIGARs, events; affine scores are sums of the same doubles in the
same order, so they are compared for equality as well (north_star allows 1e-5)."""
import os

import numpy as np
import pytest

import common
from whisper_amd import api

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    codes, packed = common.genome()
    c = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    yield c
    c.destroy()


DIST_CASES = [
    # m, k, W, variant, encoding, last_symbol
    (150, 12, None, 0, 0, -1),      # S1: mapping stage, M256, 3 words
    (150, 12, None, 0, 1, 0),       # post-processing flavour: ASCII, stale last_symbol 0
    (150, 12, None, 0, 1, 3),
    (250, 17, None, 0, 0, -1),      # S5: 4 words
    (250, 7, None, 0, 0, -1),
    (100, 10, None, 0, 0, -1),      # S7: M128
    (64, 6, None, 0, 0, -1),
    (65, 6, None, 0, 0, -1),
    (128, 8, None, 0, 0, -1),
    (192, 8, None, 0, 0, -1),
    (193, 8, None, 0, 0, -1),
    (255, 10, None, 0, 0, -1),
    (256, 10, None, 0, 0, -1),      # first M64 length
    (300, 20, None, 0, 0, -1),      # S7: M64, 5 words
    (600, 30, None, 0, 0, -1),      # 10 words
    (150, 12, None, 1, 0, -1),      # M64 forced on a short read
    (150, 26, 660, 0, 1, 0),        # S3: mate rescue window
    (30, 3, None, 0, 0, -1),
]


@pytest.mark.parametrize("m,k,W,variant,enc,last", DIST_CASES)
def test_distance(ctx, oracle, m, k, W, variant, enc, last):
    codes, packed, (blob, off, lens), tasks = common.distance_workload(m, k, 6000, seed=100 + m + k,
                                                                      encoding=enc, W=W, variant=variant)
    rc, want = oracle.distance(packed, blob, off, lens, enc, tasks, last)
    assert rc == 0
    reads = ctx.upload_reads(blob, off, lens, enc)
    got = ctx.dynamic_programming(reads, tasks, last)
    reads.free()
    assert (want["edit_distance"] <= k).sum() > 1000      # the workload has accepted candidates
    np.testing.assert_array_equal(got["edit_distance"], want["edit_distance"])
    np.testing.assert_array_equal(got["pos"], want["pos"])


def test_distance_mixed_lengths_and_empty(ctx, oracle):
    import torch
    codes, packed = common.genome()
    g = torch.Generator().manual_seed(7)
    lens_ = [1, 2, 63, 64, 65, 127, 128, 129, 150, 151, 191, 192, 193, 255]
    reads_codes, tasks = [], []
    for i, m in enumerate(lens_ * 20):
        p = int(torch.randint(20000, 900000, (1,), generator=g))
        r = codes[p:p + m].clone()
        if m > 4:
            r[int(torch.randint(0, m, (1,), generator=g))] = int(torch.randint(0, 4, (1,), generator=g))
        r[r > 3] = 4
        reads_codes.append(r)
        k = min(5, m)
        tasks.append((i, p - k, m + 2 * k, k, 0, 0))
    blob, off, lens = common.mixed_reads(reads_codes, 0)
    tasks = np.array(tasks, api.DIST_TASK)
    rc, want = oracle.distance(packed, blob, off, lens, 0, tasks, -1)
    reads = ctx.upload_reads(blob, off, lens, 0)
    got = ctx.dynamic_programming(reads, tasks, -1)
    np.testing.assert_array_equal(got, want)
    # empty batch is a no-op
    assert len(ctx.dynamic_programming(reads, tasks[:0], -1)) == 0
    reads.free()


def test_distance_errors(ctx):
    codes, packed, (blob, off, lens), tasks = common.distance_workload(150, 12, 300, seed=3)
    reads = ctx.upload_reads(blob, off, lens, 0)
    bad = tasks.copy()
    bad["max_distance_in_ref"][5] = 0                  # the reference throws runtime_error
    with pytest.raises(api.WamdError) as e:
        ctx.dynamic_programming(reads, bad)
    assert e.value.status == 5
    bad = tasks.copy()
    bad["max_distance_in_ref"][7] = 20000              # above max_text_len (16384)
    with pytest.raises(api.WamdError) as e:
        ctx.dynamic_programming(reads, bad)
    assert e.value.status == 5
    bad = tasks.copy()
    bad["ref_pos"][9] = codes.numel() - 10             # window leaves the genome
    with pytest.raises(api.WamdError) as e:
        ctx.dynamic_programming(reads, bad)
    assert e.value.status == 6
    # the context stays usable
    got = ctx.dynamic_programming(reads, tasks)
    assert (got["edit_distance"] != 0xFFFFFFFF).all()
    reads.free()


@pytest.mark.parametrize("k", [1, 2, 3, 5, 6])
def test_levdiag_stage_segments(ctx, oracle, k):
    for j, (fl, fr) in enumerate(common.stage_segments(150, k)):
        codes, packed, (blob, off, lens), tasks = common.levdiag_workload(150, k, fl, fr, 3000,
                                                                         seed=200 + 10 * k + j)
        rc, want = oracle.levdiag(packed, codes.numel(), 0, blob, off, lens, tasks)
        reads = ctx.upload_reads(blob, off, lens, 0, orientations=1)
        got = ctx.lev_diag(reads, tasks)
        reads.free()
        assert (want <= k).sum() > 300
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("m,ks", [(250, (1, 2, 3, 4)), (200, (6,)), (300, (4, 6)), (512, (5,)), (600, (3, 6)),
                                  (1000, (6,))])
def test_levdiag_long_reads(ctx, oracle, m, ks):
    """Reads above 160 symbols: levdiag_kernel<34,128> (161..256; BASELINE configs[4] runs LevDiag in
    stages 1-4 at 250 bp), <66,64> (257..512) and <130,64> (513..1024), every stage segment."""
    for k in ks:
        for j, (fl, fr) in enumerate(common.stage_segments(m, k)):
            codes, packed, (blob, off, lens), tasks = common.levdiag_workload(m, k, fl, fr, 1500,
                                                                             seed=1200 + m + 10 * k + j)
            rc, want = oracle.levdiag(packed, codes.numel(), 0, blob, off, lens, tasks)
            reads = ctx.upload_reads(blob, off, lens, 0, orientations=1)
            got = ctx.lev_diag(reads, tasks)
            reads.free()
            assert (want <= k).sum() > 150
            np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("frac,k", [(0.1, 5), (0.02, 3), (0.97, 6)])
def test_levdiag_sparse_and_dense_batches(ctx, oracle, frac, k):
    """Few related candidates among loci that only share the anchor (the prefix side ends most of them and the block
    packs the rest between the sides), and nearly all related (nothing moves)."""
    import torch
    from whisper_amd import synth
    codes, packed = common.genome()
    for j, (fl, fr) in enumerate(common.stage_segments(150, k)):
        rd = synth.make_reads(codes, 5000, 150, k, 1700 + 10 * k + j, rc_frac=0.0, protect=(fl, fr), true_frac=frac)
        blob, off, lens = common.host_reads(rd, 0)
        tasks = synth.levdiag_tasks(rd, k, fl, fr, 0)
        rc, want = oracle.levdiag(packed, codes.numel(), 0, blob, off, lens, tasks)
        reads = ctx.upload_reads(blob, off, lens, 0, orientations=1)
        got = ctx.lev_diag(reads, tasks)
        reads.free()
        assert (want <= k).sum() > 20
        np.testing.assert_array_equal(got, want)


def test_levdiag_mixed_lengths(ctx, oracle):
    """One batch holding reads of 40 .. 250 symbols: the kernel is picked by the longest read."""
    import torch
    codes, packed = common.genome()
    g = torch.Generator().manual_seed(17)
    reads_codes, tasks = [], []
    for i in range(3000):
        m = [40, 77, 100, 150, 161, 200, 249, 250][i % 8]
        k = 1 + i % 6
        fl, fr = common.stage_segments(m, k)[i % (k + 1)]
        p = int(torch.randint(20000, 900000, (1,), generator=g))
        r = codes[p:p + m].clone()
        for _ in range(int(torch.randint(0, k + 2, (1,), generator=g))):
            z = int(torch.randint(0, m, (1,), generator=g))
            if not (fl <= z < fr):
                r[z] = (int(r[z]) + 1) & 3
        r[r > 3] = 4
        reads_codes.append(r)
        tasks.append((i, p, fl, fr, k, 1))
    blob, off, lens = common.mixed_reads(reads_codes, 0)
    tasks = np.array(tasks, api.LEVDIAG_TASK)
    rc, want = oracle.levdiag(packed, codes.numel(), 0, blob, off, lens, tasks)
    reads = ctx.upload_reads(blob, off, lens, 0, orientations=1)
    got = ctx.lev_diag(reads, tasks)
    reads.free()
    assert (want <= tasks["max_mismatches"]).sum() > 800
    np.testing.assert_array_equal(got, want)


def test_levdiag_odd_shapes(ctx, oracle):
    """What the stage segments never produce, task by task in one batch: anchors shorter than k (down to empty: the
    banded kernel's phantom-row build), no prefix / no suffix, sides of k rows or fewer, k = 0, loci on the genome's
    N runs with reads that carry N as 4 (never equal to the genome's 5) or as 5 (equal: the lanes the banded kernel
    hands to the run-extension kernel)."""
    codes, packed, (blob, off, lens), tasks = common.levdiag_odd_workload(6000, 4242)
    rc, want = oracle.levdiag(packed, codes.numel(), 0, blob, off, lens, tasks)
    assert rc == 0
    reads = ctx.upload_reads(blob, off, lens, 0, orientations=1)
    got = ctx.lev_diag(reads, tasks)
    reads.free()
    ok = want <= tasks["max_mismatches"]
    assert ok.sum() > 1500 and (~ok).sum() > 500
    on_n = np.arange(len(tasks)) % 4 == 0
    assert (ok & on_n).sum() > 100
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("k", [2, 4, 6])
def test_levdiag_low_complexity(oracle, k):
    """Tandem repeats: many equally good alignments, anchors that extend far, runs that end
    on the band's edge diagonals."""
    gen = common.low_complexity_genome()
    codes, packed = gen
    c = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    try:
        for j, (fl, fr) in enumerate(common.stage_segments(120, k)):
            _, _, (blob, off, lens), tasks = common.levdiag_workload(120, k, fl, fr, 4000, seed=900 + 10 * k + j,
                                                                    gen=gen)
            rc, want = oracle.levdiag(packed, codes.numel(), 0, blob, off, lens, tasks)
            reads = c.upload_reads(blob, off, lens, 0, orientations=1)
            got = c.lev_diag(reads, tasks)
            reads.free()
            assert (want <= k).sum() > 300
            np.testing.assert_array_equal(got, want)
    finally:
        c.destroy()


@pytest.mark.parametrize("k,cur_offset,min_len", [(5, 0, 150), (1, 0, 150), (6, 17, 150), (7, 0, 250), (3, 62, 250),
                                                   (4, 9, 100), (8, 33, 300)])
def test_cf_filter(ctx, oracle, k, cur_offset, min_len):
    """Row C: dinucleotide histograms, both strands, odd / even window starts, ragged read lengths."""
    codes, packed, (blob, off, lens), tasks = common.cf_workload(6000, seed=60 + k, k=k, cur_offset=cur_offset,
                                                                 min_len=min_len)
    rc, want = oracle.cf_filter(packed, codes.numel(), cur_offset, blob, off, lens, tasks)
    assert rc == 0
    ctx.set_reference_params(codes.numel(), cur_offset)
    try:
        reads = ctx.upload_reads(blob, off, lens, 0, orientations=1)
        got = ctx.cf_filter(reads, tasks)
        bad = tasks[:8].copy()
        bad["fixed_left"] = 1                      # one-symbol region: the reference's walk underflows
        with pytest.raises(api.WamdError):
            ctx.cf_filter(reads, bad)
        reads.free()
    finally:
        ctx.set_reference_params(codes.numel(), 0)
    assert 500 < (want <= 4 * k).sum() < 5800
    np.testing.assert_array_equal(got, want)


def test_levdiag_rc_and_offset(ctx, oracle):
    """dir_gen_flag = 0 window arithmetic (edit_dist.cpp:114-115) and a non-zero cur_offset."""
    codes, packed, (blob, off, lens), tasks = common.levdiag_workload(100, 4, 40, 60, 3000, seed=77,
                                                                     cur_offset=40)
    ref_size = codes.numel()
    t2 = tasks.copy()
    half = len(t2) // 2
    # rc flavour: g0 = ref_size - position - m + cur_offset - k  ==  p - k
    p = tasks["position_in_text"][:half].astype(np.int64) - 40
    t2["position_in_text"][:half] = (ref_size - p - 100 + 40).astype(np.uint32)
    t2["dir_gen_flag"][:half] = 0
    ctx.set_reference_params(ref_size, 40)
    try:
        rc, want = oracle.levdiag(packed, ref_size, 40, blob, off, lens, t2)
        reads = ctx.upload_reads(blob, off, lens, 0, orientations=1)
        got = ctx.lev_diag(reads, t2)
        reads.free()
    finally:
        ctx.set_reference_params(ref_size, 0)
    assert (want <= 4).sum() > 500
    np.testing.assert_array_equal(got, want)


ALIGN_CASES = [
    # m, k, W, last_symbol, refine, variant
    (150, 12, None, 0, 1, 0),
    (150, 12, None, 0, 0, 0),
    (150, 13, 177, 0, 1, 0),        # S4: fill_cigar_with_lev
    (150, 26, 660, 0, 1, 0),        # S3: mate rescue
    (250, 17, None, 0, 1, 0),
    (250, 44, 700, 0, 1, 0),        # S6: ed above 30 -> 128-row band
    (100, 10, None, 0, 1, 0),
    (300, 20, None, 0, 1, 0),
    (150, 12, None, 3, 1, 0),
    (150, 12, None, 0, 1, 1),       # M64 on a short read
    (60, 40, None, 0, 1, 0),        # ed > 30 with a single word: whole columns kept
    (150, 26, 1427, 0, 1, 0),       # rescue window before the insert-size model is trained (sam.cpp:185-188)
    (250, 44, 1588, 0, 1, 0),       # the same at 250 bp: 500 + 7 * 125 + 2 * 44 ... and beyond
    (150, 26, 9000, 0, 0, 0),       # a window as long as the genome's N border allows
    (700, 20, None, 0, 1, 0),       # reads too long for the replay's LDS rows: its form that writes straight to memory
]


def test_align_round_trip_at_scale(ctx, oracle):
    """Size-independent property at 2^17 tasks (the oracle does a 4096-task sample): every
    ext-CIGAR, replayed over its read and genome window, consumes the whole read, ends at
    end_pos, names the right bases and holds exactly edit_distance (+1 when refined) edits."""
    n = 1 << 17
    codes, packed, (blob, off, lens), tasks = common.distance_workload(150, 13, n, seed=4242, encoding=1,
                                                                      W=177, true_frac=0.9)
    reads = ctx.upload_reads(blob, off, lens, 1)
    res, cig = ctx.align(reads, tasks, last_symbol=0, refine=1, cigar_stride=512)
    reads.free()
    assert res["ok"].sum() > n // 2
    fail = oracle.check_ext_cigars(packed, blob, off, lens, tasks, res, cig)
    bad = np.nonzero(fail)[0]
    assert len(bad) == 0, "task %d fails the replay with code %d" % (bad[0], fail[bad[0]])
    sel = np.arange(0, n, n // 4096)
    rc, want, wcig = oracle.align(packed, blob, off, lens, 1, tasks[sel], last_symbol=0, refine=1, cigar_stride=512)
    assert rc == 0
    for f in ("ok", "edit_distance", "end_pos", "begin_pos", "num_events", "cigar_len", "refined", "affine_score"):
        np.testing.assert_array_equal(res[f][sel], want[f], err_msg=f)
    for i, j in enumerate(sel):
        ln = int(want["cigar_len"][i]) + 1
        assert bytes(cig[j, :ln]) == bytes(wcig[i, :ln])


@pytest.mark.parametrize("m,k,W,last,refine,variant", ALIGN_CASES)
def test_align(ctx, oracle, m, k, W, last, refine, variant):
    codes, packed, (blob, off, lens), tasks = common.distance_workload(m, k, 4000, seed=300 + m + k,
                                                                      encoding=1, W=W, variant=variant)
    stride = 4 * ((2 * (m + k) + 8 + 3) // 4)
    rc, want, wcig = oracle.align(packed, blob, off, lens, 1, tasks, last, refine, cigar_stride=stride)
    assert rc == 0
    reads = ctx.upload_reads(blob, off, lens, 1)
    got, gcig = ctx.align(reads, tasks, last, refine, cigar_stride=stride)
    reads.free()
    assert want["ok"].sum() > 500
    for f in want.dtype.names:
        np.testing.assert_array_equal(got[f], want[f], err_msg=f)
    for i in np.nonzero(want["ok"])[0]:
        n = want["cigar_len"][i]
        assert bytes(gcig[i, :n + 1]) == bytes(wcig[i, :n + 1]), i


def test_align_dev_workspace_sizes(ctx, oracle):
    """wamd_align_dev on a mate-rescue batch (ed up to 26: the 64-row and wider classes) with the workspace of
    wamd_align_workspace_bytes (sized for the banded class), of wamd_align_workspace_bytes_for(max_ed) and with
    twice that: the same answers - the size only decides how many blocks of a wide class are resident -
    and a workspace below the smallest is an error, not a wild write."""
    import torch
    dev = torch.device("cuda", 0)
    m, k, W, n = 150, 26, 660, 6000
    codes, packed, (blob, off, lens), tasks = common.distance_workload(m, k, n, seed=977, encoding=1, W=W)
    stride = 4 * ((2 * (m + k) + 8 + 3) // 4)
    rc, want, wcig = oracle.align(packed, blob, off, lens, 1, tasks, 0, 1, cigar_stride=stride)
    assert rc == 0 and (want["edit_distance"][want["ok"] == 1] > 14).sum() > 100
    reads = ctx.upload_reads(blob, off, lens, 1)
    d_t = torch.from_numpy(np.ascontiguousarray(tasks).view(np.uint8).reshape(-1).copy()).to(dev)
    small = ctx.align_workspace_bytes(reads, n)
    wide = ctx.align_workspace_bytes(reads, n, max_ed=k)
    assert wide >= small
    for wsb in (small, wide, 2 * wide):
        d_r = torch.zeros(n * api.ALIGN_RESULT.itemsize, dtype=torch.uint8, device=dev)
        d_c = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
        d_w = torch.zeros(wsb + 256, dtype=torch.uint8, device=dev)
        pad = (-d_w.data_ptr()) % 256
        ctx.align_dev(reads, d_t, n, d_r, d_c, stride, d_w[pad:], wsb, last_symbol=0, refine=1)
        ctx.synchronize()
        assert ctx.take_flags() == 0
        got = d_r.cpu().numpy().view(api.ALIGN_RESULT)
        gcig = d_c.cpu().numpy().reshape(n, stride)
        for f in want.dtype.names:
            np.testing.assert_array_equal(got[f], want[f], err_msg="%s at workspace %d" % (f, wsb))
        for i in np.nonzero(want["ok"])[0][::7]:
            ln = want["cigar_len"][i]
            assert bytes(gcig[i, :ln + 1]) == bytes(wcig[i, :ln + 1]), i
    with pytest.raises(api.WamdError):
        ctx.align_dev(reads, d_t, n, d_r, d_c, stride, d_w[pad:], 4096, last_symbol=0, refine=1)
    reads.free()


@pytest.mark.parametrize("m,enc,mcf,stage,max_indel", [(150, 0, 1.0, 5, 50), (150, 1, 0.6, 5, 50), (100, 0, 0.5, 4, 50),
                                                        (250, 0, 1.0, 4, 50),
                                                        (300, 0, 0.6, 8, 50),      # reads above 256: indel_match_kernel<false>
                                                        (150, 0, 0.6, 5, 100),     # max_indel_size above 80: the same
                                                        (150, 0, 0.6, 5, 80),      # the staged kernel's largest
                                                        (600, 1, 0.7, 8, 120)])
def test_indel_match_chain(ctx, oracle, m, enc, mcf, stage, max_indel):
    """Match -> GetExtCigarNew -> serialize, each against the oracle."""
    from oracle import bindings
    codes, packed, (blob, off, lens), tasks = common.indel_workload(2000, m, seed=400 + m + enc + max_indel, encoding=enc,
                                                                   stage_major=stage, max_indel=max_indel,
                                                                   max_mm=12 if m <= 250 else 20)
    _, want = oracle.indel_match(packed, blob, off, lens, enc, tasks, bindings.default_indel_params(mcf))
    reads = ctx.upload_reads(blob, off, lens, enc, orientations=1)
    got = ctx.indel_match(reads, tasks, api.default_indel_params(mcf))
    assert want["found"].sum() > 500
    np.testing.assert_array_equal(got, want)
    ct = common.indel_cigar_tasks(tasks, want)
    stride = 4 * ((2 * (m + 2 * max_indel + 20) + 8 + 3) // 4)
    _, wres, wcig = oracle.indel_ext_cigar(packed, blob, off, lens, enc, ct, cigar_stride=stride)
    gres, gcig = ctx.indel_ext_cigar(reads, ct, cigar_stride=stride)
    reads.free()
    np.testing.assert_array_equal(gres, wres)
    for i in range(len(wres)):
        n = wres["cigar_len"][i] + 1
        assert bytes(gcig[i, :n]) == bytes(wcig[i, :n]), i
    d = (np.arange(len(want)) % 2).astype(np.uint8)
    _, wwire, wsz = oracle.serialize(want, d)
    gwire, gsz = ctx.serialize_mappings(want, d)
    np.testing.assert_array_equal(gsz, wsz)
    np.testing.assert_array_equal(gwire, wwire)


def test_indel_match_rev_comp(ctx, oracle):
    """rev_comp orientation of the read (Preprocess with genome_t::rev_comp, indels.cpp:969-971)."""
    import torch
    from whisper_amd import synth
    codes, packed, (blob, off, lens), tasks = common.indel_workload(1500, 150, seed=8)
    # store every read reverse-complemented and ask for the rev_comp orientation
    n = len(tasks)
    stride = blob.size // n
    rd = torch.zeros((n, 150), dtype=torch.uint8)
    b = torch.from_numpy(blob.reshape(n, stride))
    rd[:, 0::2] = b >> 4
    rd[:, 1::2] = b & 15
    rc = synth.rev_comp(rd)
    blob2, off2, lens2 = common.host_reads(dict(codes=rc), 0)
    t2 = tasks.copy()
    t2["flags"] = 1
    _, want = oracle.indel_match(packed, blob2, off2, lens2, 0, t2)
    _, direct = oracle.indel_match(packed, blob, off, lens, 0, tasks)
    np.testing.assert_array_equal(want, direct)
    reads = ctx.upload_reads(blob2, off2, lens2, 0, orientations=2)
    got = ctx.indel_match(reads, t2)
    reads.free()
    np.testing.assert_array_equal(got, want)


def test_clip_boundary(ctx, oracle):
    rec, cig = common.clip_workload(3000, 150, seed=21)
    _, wrec, wcig = oracle.clip_boundary(rec, cig)
    grec, gcig = ctx.clip_boundary(rec, cig)
    assert (wrec["no_clippings"] > 0).sum() > 1000
    np.testing.assert_array_equal(grec, wrec)
    for i in range(len(rec)):
        n = wrec["cigar_len"][i] + 1
        assert bytes(gcig[i, :n]) == bytes(wcig[i, :n]), i


def _same_clip_output(grec, gcig, wrec, wcig):
    for f in wrec.dtype.names:
        if f != "reserved":
            np.testing.assert_array_equal(grec[f], wrec[f], err_msg=f)
    for i in range(len(wrec)):
        n = wrec["cigar_len"][i] + 1
        assert bytes(gcig[i, :n]) == bytes(wcig[i, :n]), i


def test_clip_illegal(ctx, oracle):
    """Row L: clipIllegalPositions - mappings hanging over either end of their chromosome."""
    t, rec, cig = common.clip_illegal_workload(oracle, 4000, seed=33)
    rc, wrec, wcig = oracle.clip_illegal(t, rec, cig)
    assert rc == 0
    grec, gcig = ctx.clip_illegal(t, rec, cig)
    assert (wrec["no_clippings"] > 0).sum() > 2000
    _same_clip_output(grec, gcig, wrec, wcig)
    # a record that claims more read than its ext-CIGAR holds is an error, not a wild write
    t2, rec2, cig2 = t[:64].copy(), rec[:64].copy(), cig[:64, :160].copy()
    t2["read_length"] = 400
    rec2["ref_seq_pos"] = -3
    with pytest.raises(api.WamdError):
        ctx.clip_illegal(t2, rec2, cig2)


def test_clip_overlapping_pairs(ctx, oracle):
    """Row L: clipOverlappingPairs with clipLeft / clipRight."""
    t, rec, cig = common.clip_pairs_workload(oracle, 4000, seed=34)
    rc, wrec, wcig = oracle.clip_overlapping_pairs(t, rec, cig)
    assert rc == 0
    grec, gcig = ctx.clip_overlapping_pairs(t, rec, cig)
    assert (wrec["no_clippings"] > 0).sum() > 2000
    _same_clip_output(grec, gcig, wrec, wcig)
    # ref_length larger than the ext-CIGAR spans: the reference would run off the string
    t2, rec2, cig2 = t[:32].copy(), rec[:64].copy(), cig[:64].copy()
    t2["ref_seq_id"] = 1
    t2["direct"][:, 0] = 0
    t2["direct"][:, 1] = 1
    t2["perfect"] = 0
    rec2["left_clipping"] = 0
    rec2["right_clipping"] = 0
    rec2["ref_seq_pos"][0::2] = 100
    rec2["ref_seq_pos"][1::2] = 900
    rec2["ref_length"][0::2] = 1000
    with pytest.raises(api.WamdError):
        ctx.clip_overlapping_pairs(t2, rec2, cig2)


@pytest.mark.parametrize("lowcx,fresh,enc,W", [(0, 0, 1, 700), (0, 1, 1, 1427), (1, 0, 1, 700), (1, 1, 0, 260),
                                                (1, 0, 1, 4000)])
def test_softclip_match(ctx, oracle, lowcx, fresh, enc, W):
    """Row S: the hashing mate rescue, CSoftClipping::{preprocess*, match, getExtCigar}."""
    from oracle import bindings as ob
    gen = common.low_complexity_genome() if lowcx else common.genome()
    codes, packed, (blob, off, lens), tasks = common.softclip_workload(2500, seed=50 + W + fresh, W=W, encoding=enc,
                                                                       gen=gen, lens_jitter=bool(fresh))
    _, want, wcig = oracle.softclip_match(packed, blob, off, lens, enc, tasks, ob.default_softclip_params(fresh_object=fresh))
    c2 = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    try:
        reads = c2.upload_reads(blob, off, lens, enc, orientations=3)
        got, gcig = c2.softclip_match(reads, tasks, api.default_softclip_params(fresh_object=fresh))
        reads.free()
    finally:
        c2.destroy()
    assert want["found"].sum() > 1500 and (want["del_size"] != 0).sum() > 100
    np.testing.assert_array_equal(got, want)
    for i in range(len(want)):
        n = want["cigar_len"][i] + 1
        assert bytes(gcig[i, :n]) == bytes(wcig[i, :n]), i


@pytest.mark.parametrize("m,W,mql", [(700, 2000, 1024), (1000, 3000, 1024), (400, 1200, 700), (650, 12000, 655)])
def test_softclip_match_long_reads(oracle, m, W, mql):
    """Reads of 700 and 1 000 symbols (tables of 32 768 and 65 536 cells, two-byte read positions): the sizes every
    other entry point takes.  Low-complexity text as well, where chains are long and the queue is served in passes; the last
    case leaves room for two windows per wave instead of four."""
    from oracle import bindings as ob
    for gen in (common.genome(), common.low_complexity_genome()):
        codes, packed, (blob, off, lens), tasks = common.softclip_workload(300, seed=m + W, m=m, W=W, encoding=1, gen=gen,
                                                                           lens_jitter=True)
        _, want, wcig = oracle.softclip_match(packed, blob, off, lens, 1, tasks,
                                              ob.default_softclip_params(max_query_len=mql, fresh_object=0), cigar_stride=1536)
        c2 = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
        try:
            reads = c2.upload_reads(blob, off, lens, 1, orientations=3)
            got, gcig = c2.softclip_match(reads, tasks, api.default_softclip_params(max_query_len=mql, fresh_object=0),
                                          cigar_stride=1536)
            reads.free()
        finally:
            c2.destroy()
        assert want["found"].sum() > 150
        np.testing.assert_array_equal(got, want)
        for i in range(len(want)):
            n = want["cigar_len"][i] + 1
            assert bytes(gcig[i, :n]) == bytes(wcig[i, :n]), i


def test_empty_batches(ctx):
    """Every host-pointer entry point with no tasks (an empty bin, a pair block with no rescue to do) is a no-op that
    returns empty results and leaves no error behind."""
    codes, packed, (blob, off, lens), tasks = common.softclip_workload(8, seed=1, W=300)
    reads = ctx.upload_reads(blob, off, lens, 1, orientations=3)
    z = lambda dt: np.zeros(0, dt)
    assert len(ctx.dynamic_programming(reads, z(api.DIST_TASK), -1)) == 0
    assert len(ctx.lev_diag(reads, z(api.LEVDIAG_TASK))) == 0
    assert len(ctx.cf_filter(reads, z(api.CF_TASK))) == 0
    res, cig = ctx.align(reads, z(api.DIST_TASK))
    assert len(res) == 0 and cig.shape[0] == 0
    assert len(ctx.indel_match(reads, z(api.INDEL_TASK))) == 0
    res, cig = ctx.indel_ext_cigar(reads, z(api.INDEL_CIGAR_TASK))
    assert len(res) == 0
    res, cig = ctx.softclip_match(reads, tasks[:0])
    assert len(res) == 0
    res, ext = ctx.rescue_mates(reads, z(api.RESCUE_TASK))
    assert len(res) == 0
    out = ctx.finish_pairs(reads, z(api.FINISH_TASK))
    assert len(out[0]) == 0
    rec, cg = ctx.clip_boundary(z(api.CLIP_RECORD), np.zeros((0, 64), np.uint8))
    assert len(rec) == 0
    rec, cg = ctx.clip_illegal(z(api.CLIP_ILLEGAL_TASK), z(api.CLIP_RECORD), np.zeros((0, 64), np.uint8))
    assert len(rec) == 0
    rec, cg = ctx.clip_overlapping_pairs(z(api.CLIP_PAIR_TASK), z(api.CLIP_RECORD), np.zeros((0, 64), np.uint8))
    assert len(rec) == 0
    out = ctx.sam_cigar(np.zeros((0, 64), np.uint8))
    assert len(out[0]) == 0
    err = ctx.verify_sa_candidates(reads, api.sa_batch(len(lens), 5, (50, 75), 0, api.SA_VERIFIER_DISTANCE),
                                   np.zeros(2 * len(lens), np.uint16), z(np.uint32))
    assert len(err[0] if isinstance(err, tuple) else err) == 0
    wire, sizes = ctx.serialize_mappings(z(api.MAPPING), z(np.uint8))
    assert len(sizes) == 0
    assert ctx.take_flags() == 0
    reads.free()


def test_softclip_match_errors(ctx, oracle):
    """Reads shorter than one 20-mer and windows above max_text_len are errors, as is a table of more than 65 536 cells."""
    codes, packed, (blob, off, lens), tasks = common.softclip_workload(64, seed=3, W=300)
    reads = ctx.upload_reads(blob, off, lens, 1, orientations=3)
    bad = tasks.copy()
    bad["max_distance_in_ref"] = 20000
    with pytest.raises(api.WamdError):
        ctx.softclip_match(reads, bad)
    with pytest.raises(api.WamdError):
        ctx.softclip_match(reads, tasks, api.default_softclip_params(max_query_len=4000))
    reads.free()
    short = np.frombuffer(b"ACGTACGTAC", np.uint8).copy()
    r2 = ctx.upload_reads(short, np.array([0], np.uint64), np.array([10], np.uint32), 1, orientations=3)
    with pytest.raises(api.WamdError):
        ctx.softclip_match(r2, tasks[:1], api.default_softclip_params())
    r2.free()


def test_sam_cigar(ctx, oracle):
    """Row X: SAM CIGAR text, BAM operations and MD:Z from ext-CIGARs of every flavour."""
    ext = common.ext_cigar_pool(oracle, n=6000, seed=21)
    rc, wc, wm, wb, wl = oracle.sam_cigar(ext)
    gc, gm, gb, gl = ctx.sam_cigar(ext)
    np.testing.assert_array_equal(gl, wl)
    for i in range(len(ext)):
        assert bytes(gc[i, :wl["cigar_len"][i] + 1]) == bytes(wc[i, :wl["cigar_len"][i] + 1])
        assert bytes(gm[i, :wl["mdz_len"][i] + 1]) == bytes(wm[i, :wl["mdz_len"][i] + 1])
        nb = wl["cigar_bin_len"][i]
        np.testing.assert_array_equal(gb[i, :nb], wb[i, :nb])
    with pytest.raises(api.WamdError):             # outputs that cannot fit are an error, not a truncation
        ctx.sam_cigar(ext[:64], out_stride=16)


def _restride(cig, stride):
    out = np.zeros((cig.shape[0], stride), np.uint8)
    w = min(stride, cig.shape[1])
    out[:, :w] = cig[:, :w]
    return out


@pytest.mark.parametrize("m,stride,n", [(150, 512, 2999), (150, 400, 1537), (250, 640, 1001), (100, 192, 777),
                                        (150, 388, 1000)])
def test_post_rows_slot_layouts(ctx, oracle, m, stride, n):
    """Rows L / B / X through their stand-alone entry points on every way their kernel stages the slots: strings a
    lane keeps in registers (strides that are multiples of 16: measured in registers at 150 bp, by the wave above 192
    bytes at 250 bp, whole slots at stride 192), the copy loops (stride 388), a ragged last wave and an odd count."""
    t, rec, cig = common.clip_illegal_workload(oracle, n, seed=60 + m + stride % 7, m=m)
    cig = _restride(cig, stride)
    rc, wrec, wcig = oracle.clip_illegal(t, rec, cig)
    assert rc == 0
    grec, gcig = ctx.clip_illegal(t, rec, cig)
    _same_clip_output(grec, gcig, wrec, wcig)
    _, wrec2, wcig2 = oracle.clip_boundary(wrec, wcig)
    grec2, gcig2 = ctx.clip_boundary(grec, gcig)
    np.testing.assert_array_equal(grec2, wrec2)
    for i in range(len(wrec2)):
        k = wrec2["cigar_len"][i] + 1
        assert bytes(gcig2[i, :k]) == bytes(wcig2[i, :k]), i
    rc, wc, wm, wb, wl = oracle.sam_cigar(wcig2)
    gc, gm, gb, gl = ctx.sam_cigar(gcig2)
    np.testing.assert_array_equal(gl, wl)
    for i in range(len(wl)):
        assert bytes(gc[i, :wl["cigar_len"][i] + 1]) == bytes(wc[i, :wl["cigar_len"][i] + 1])
        assert bytes(gm[i, :wl["mdz_len"][i] + 1]) == bytes(wm[i, :wl["mdz_len"][i] + 1])
    tp, recp, cigp = common.clip_pairs_workload(oracle, n // 2, seed=61 + m, m=m)
    cigp = _restride(cigp, stride)
    rc, wrecp, wcigp = oracle.clip_overlapping_pairs(tp, recp, cigp)
    assert rc == 0
    grecp, gcigp = ctx.clip_overlapping_pairs(tp, recp, cigp)
    _same_clip_output(grecp, gcigp, wrecp, wcigp)


@pytest.mark.parametrize("seed,m,flank", [(31, 150, 6), (32, 250, 8), (33, 100, 5), (34, 150, 13)])
def test_finish_pairs_chain(ctx, oracle, seed, m, flank):
    """Row N1's stand-in: the reference's post-processing order on one set of pairs - fill_cigar (perfect / lev /
    indel-matcher records), clipIllegalPositions, clipBoundaryPositions, clipOverlappingPairs,
    extract_cigar_and_mdz - through one entry point; every SAM-visible field and string against the oracle."""
    (codes, packed), (blob, off, lens), desc, ridx, maps, direct, distant, stale = common.finish_workload(
        oracle, 1500, seed, m=m, flank=flank)
    tasks = common.finish_tasks(desc, ridx, lens, maps, direct, distant, stale)
    from oracle import bindings as ob
    rc, *want = oracle.finish_pairs(packed, blob, off, lens, tasks, ob.default_finish_params(flank), ext_stride=768)
    assert rc == 0 and want[0]["ok"].all()
    reads = ctx.upload_reads(blob, off, lens, 1, orientations=3)
    got = ctx.finish_pairs(reads, tasks, api.default_finish_params(flank), ext_stride=768)
    common.same_finish_output(got, want)
    # the switches of CParams: no refinement, no boundary clipping, no paired clipping
    for sw in ((0, 1, 1), (1, 0, 1), (1, 1, 0)):
        rc, *want = oracle.finish_pairs(packed, blob, off, lens, tasks[:600], ob.FinishParams(flank, *sw), ext_stride=768)
        got = ctx.finish_pairs(reads, tasks[:600], api.FinishParams(flank, *sw), ext_stride=768)
        common.same_finish_output(got, want)
    # the fused passes without the per-slot length hints: every wave serves its slots in two halves at the full stride
    # (what it also does when the packed strings of a wave do not fit its LDS)
    rc, *want = oracle.finish_pairs(packed, blob, off, lens, tasks, ob.default_finish_params(flank), ext_stride=768)
    ctx.set_test_hooks(1)
    try:
        got = ctx.finish_pairs(reads, tasks, api.default_finish_params(flank), ext_stride=768)
    finally:
        ctx.set_test_hooks(0)
    common.same_finish_output(got, want)
    reads.free()


@pytest.mark.parametrize("seed,m", [(43, 150), (44, 250)])
def test_finish_pairs_distant_from_pair_and_failing_reads(ctx, oracle, seed, m):
    """WAMD_DISTANT_FROM_PAIR: isDistant of sam.cpp:1881-1884 computed on the device - for the second read of a pair
    from the first one's position and length after ITS alignment and clipping passes - under an insert-size model
    that calls part of the pairs distant; and reads fill_cigar_with_lev cannot align: that slot alone is not
    finished, its mate is."""
    (codes, packed), (blob, off, lens), desc, ridx, maps, direct, distant, stale = common.finish_workload(
        oracle, 1500, seed, m=m, flank=7, from_pair_frac=0.7)
    lev = np.nonzero((maps["type"] == 0) & (maps["no_mismatches"] > 2))[0]
    maps["no_mismatches"][lev[::11]] = 1
    tasks = common.finish_tasks(desc, ridx, lens, maps, direct, distant, stale, isize=common.FINISH_ISIZE)
    from oracle import bindings as ob
    rc, *want = oracle.finish_pairs(packed, blob, off, lens, tasks, ob.default_finish_params(7), ext_stride=768)
    bad = want[0]["ok"] == 0
    assert rc == 0 and 20 < bad.sum() < 400 and want[0]["ok"][np.nonzero(bad)[0] ^ 1].sum() > 10
    fp = (tasks["distant"] == 2) & ~bad
    dist = np.isin(want[0]["method"], (1, 3))
    assert (fp & dist).sum() > 100 and (fp & ~dist).sum() > 100
    assert (dist[0::2] != dist[1::2])[fp[0::2] & fp[1::2]].sum() > 0    # the mate's answer follows the first read's alignment
    reads = ctx.upload_reads(blob, off, lens, 1, orientations=3)
    got = ctx.finish_pairs(reads, tasks, api.default_finish_params(7), ext_stride=768)
    reads.free()
    common.same_finish_output(got, want)


def test_finish_pairs_dev_stays_on_the_device(ctx, oracle):
    """The _dev form on device-resident tasks and slots (caller's workspace, caller's stream, nothing allocated by
    the library, no host round trip between the stages), plus a pair whose record cannot be aligned."""
    import torch
    (codes, packed), (blob, off, lens), desc, ridx, maps, direct, distant, stale = common.finish_workload(oracle, 800, 41)
    tasks = common.finish_tasks(desc, ridx, lens, maps, direct, distant, stale)
    lev = np.nonzero((tasks["type"] == 0) & (tasks["err_edit_distance"] > 2))[0]
    tasks["err_edit_distance"][lev[0]] = 0          # claims fewer errors than the read has: fill_cigar_with_lev fails
    tasks["penalty"][lev[0]] = 1
    rc, *want = oracle.finish_pairs(packed, blob, off, lens, tasks, ext_stride=512)
    assert rc == 0 and want[0]["ok"].sum() == len(tasks) - 1 and want[0]["ok"][lev[0] ^ 1] == 1   # the mate is finished
    dev = torch.device("cuda", 0)
    n = len(tasks)
    reads = ctx.upload_reads(blob, off, lens, 1, orientations=3)
    d_tasks = torch.from_numpy(tasks.view(np.uint8).copy()).to(dev)
    d_res = torch.zeros(n * api.FINISH_RESULT.itemsize, dtype=torch.uint8, device=dev)
    d_ext = torch.full((n, 512), 0x7f, dtype=torch.uint8, device=dev)
    d_cig = torch.full((n, 1040), 0x7f, dtype=torch.uint8, device=dev)
    d_mdz = torch.full((n, 1040), 0x7f, dtype=torch.uint8, device=dev)
    wsb = ctx.finish_workspace_bytes(reads, n // 2)
    d_ws = torch.zeros(wsb + 256, dtype=torch.uint8, device=dev)
    pad = (-d_ws.data_ptr()) % 256
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    ctx.finish_pairs_dev(reads, d_tasks, n // 2, d_res, d_ext, 512, d_cig, 1040, d_mdz, 1040, d_ws[pad:], wsb,
                         stream=st.cuda_stream)
    st.synchronize()
    assert ctx.take_flags(st.cuda_stream) == 0
    reads.free()
    res = d_res.cpu().numpy().view(api.FINISH_RESULT)
    common.same_finish_output((res, d_ext.cpu().numpy(), d_cig.cpu().numpy(), d_mdz.cpu().numpy()), want)


@pytest.mark.parametrize("seed,max_no_mappings,id_bytes", [(13, 1024, 5), (14, 8, 4), (15, 1, 8), (16, 3, 6), (17, 60, 3)])
def test_commit_and_result_groups(ctx, oracle, seed, max_no_mappings, id_bytes):
    """Rows K / (f)3: what rank 0 receives is what CMappingResultsCollector would hold - per-read header, lev and
    non-lev records, in the heap gatherer's order - for 20 000 reads; plus the per-read offsets and the overflow flag."""
    from oracle import bindings as ob
    t, indel, read_first, read_ids = common.commit_workload(oracle, 20000, seed)
    if id_bytes < 5:
        read_ids = read_ids & np.uint64((1 << (8 * id_bytes)) - 1)
    _, want_c = oracle.commit(t, indel)
    rc, want = oracle.result_groups(t, want_c, read_first, read_ids, ob.GroupParams(id_bytes, 4, max_no_mappings, 0))
    assert rc == 0
    committed, data, offs, flags, _ = common.gpu_result_groups(ctx, t, indel, read_first, read_ids, id_bytes, max_no_mappings)
    assert flags == 0
    np.testing.assert_array_equal(committed, want_c)
    assert len(data) == len(want) and bytes(data) == bytes(want)
    # every read with a kept mapping starts with its id at the offset reported for it
    kept = np.add.reduceat(np.concatenate([want_c["found"], [0]]), read_first[:-1].astype(np.int64)) * (np.diff(read_first) > 0)
    for r in np.nonzero(kept)[0][:200]:
        got_id = int.from_bytes(bytes(data[offs[r]:offs[r] + id_bytes]), "big")
        assert got_id == int(read_ids[r]) & ((1 << (8 * id_bytes)) - 1)
    # a buffer that cannot hold the group: flag, and nothing written past its end
    _, _, _, flags, raw = common.gpu_result_groups(ctx, t, indel, read_first, read_ids, id_bytes, max_no_mappings,
                                                   capacity=len(want) // 2)
    assert flags & 64 and len(raw) == len(want) // 2


def test_emit_lev_records(ctx, oracle):
    """Accepted candidates -> 12-byte mapping records in task order (wamd_emit_lev_records_dev)."""
    import torch
    codes, packed, (blob, off, lens), tasks = common.distance_workload(150, 12, 5000, seed=71)
    _, want = oracle.distance(packed, blob, off, lens, 0, tasks, -1)
    n = len(tasks)
    dev = torch.device("cuda", 0)
    reads = ctx.upload_reads(blob, off, lens, 0)
    d_tasks = torch.from_numpy(tasks.view(np.uint8).copy()).to(dev)
    d_res = torch.zeros(2 * n, dtype=torch.int32, device=dev)
    pos = (tasks["ref_pos"] + 12).astype(np.uint32)
    direction = (np.arange(n) % 3 != 0).astype(np.uint8)
    d_pos = torch.from_numpy(pos.view(np.int32).copy()).to(dev)
    d_dir = torch.from_numpy(direction).to(dev)
    d_rec = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    d_scr = torch.zeros(ctx.records_scratch_bytes(n), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.dynamic_programming_dev(reads, d_tasks, n, d_res)
    ctx.emit_lev_records_dev(n, d_tasks, d_res, d_pos, d_dir, d_rec, d_cnt, d_scr)
    ctx.synchronize()
    assert ctx.take_flags() == 0
    reads.free()
    ok = np.nonzero(want["edit_distance"] <= 12)[0]
    assert int(d_cnt.item()) == len(ok)
    rec = d_rec.cpu().numpy()[: len(ok)]
    # expected bytes: read_idx LE, then the reference's 5-byte lev record
    m = np.zeros(len(ok), api.MAPPING)
    m["type"] = 0
    m["pos"] = pos[ok]
    m["no_mismatches"] = want["edit_distance"][ok]
    _, wire, sizes = oracle.serialize(m, direction[ok])
    assert (sizes == 5).all()
    np.testing.assert_array_equal(rec[:, :4].copy().view("<u4")[:, 0], tasks["read_idx"][ok])
    np.testing.assert_array_equal(rec[:, 4:9], wire[:, :5])
    assert (rec[:, 9:] == 0).all()


def test_windows_at_the_genome_ends(oracle):
    """Windows that start in the genome's first dword or end in its last one (a genome without
    the N borders): distance in both modes and align; the clamped loads at either end and the
    first group of the band must see the right symbols."""
    import torch
    from whisper_amd import synth
    g = torch.Generator().manual_seed(5)
    G = 4096 + 6
    codes = torch.randint(0, 4, (G,), generator=g, dtype=torch.uint8)
    packed = synth.pack_nibbles(codes).numpy()
    m, k = 150, 12
    starts = list(range(0, 20)) + list(range(G - m - 2 * k - 1 - 20, G - m - 2 * k - 1))
    reads_codes, tasks = [], []
    for rep in range(8):
        for i, s0 in enumerate(starts):
            W = m + 2 * k
            p = s0 + int(torch.randint(0, 2 * k, (1,), generator=g))
            p = min(p, G - m)
            r = codes[p:p + m].clone()
            for _ in range(int(torch.randint(0, 6, (1,), generator=g))):
                x = int(torch.randint(0, m, (1,), generator=g))
                r[x] = (int(r[x]) + 1) & 3
            reads_codes.append(r)
            tasks.append((len(tasks), s0, W, k, 0, 0))
    tasks = np.array(tasks, api.DIST_TASK)
    c = api.Context(packed, ref_size=G, cur_offset=0)
    try:
        for enc in (0, 1):
            blob, off, lens = common.mixed_reads(reads_codes, enc)
            last = 0 if enc else -1
            rc, want = oracle.distance(packed, blob, off, lens, enc, tasks, last)
            assert rc == 0 and (want["edit_distance"] <= k).sum() > 100
            reads = c.upload_reads(blob, off, lens, enc)
            np.testing.assert_array_equal(c.dynamic_programming(reads, tasks, last), want)
            c.set_distance_mode(api.DISTANCE_ACCEPTED_ONLY)
            got = c.dynamic_programming(reads, tasks, last)
            c.set_distance_mode(api.DISTANCE_EXACT)
            rej = want["edit_distance"] > k
            np.testing.assert_array_equal(got[~rej], want[~rej])
            assert (got["edit_distance"][rej] == k + 1).all() and (got["pos"][rej] == 0).all()
            if enc:
                rc, wa, wcig = oracle.align(packed, blob, off, lens, 1, tasks, 0, 1, cigar_stride=512)
                assert rc == 0
                ga, gcig = c.align(reads, tasks, 0, 1, cigar_stride=512)
                for f in wa.dtype.names:
                    np.testing.assert_array_equal(ga[f], wa[f], err_msg=f)
                for i in np.nonzero(wa["ok"])[0]:
                    n = wa["cigar_len"][i]
                    assert bytes(gcig[i, :n + 1]) == bytes(wcig[i, :n + 1]), i
            reads.free()
    finally:
        c.destroy()


def test_reads_refill_and_stream_upload(ctx, oracle):
    """wamd_reads_refill: the next bin's reads in the buffers of an existing batch (larger and smaller than what it
    held), and wamd_reads_upload_stream: the same batch built on a caller's stream - both must verify as a fresh
    upload does."""
    import torch
    w1 = common.distance_workload(150, 12, 3000, seed=901)
    w2 = common.distance_workload(100, 10, 5000, seed=902)     # more reads, shorter
    w3 = common.distance_workload(250, 17, 700, seed=903)      # fewer reads, longer: other vector widths
    reads = ctx.upload_reads(*w1[2], api.READS_NIBBLE)
    for codes, packed, (blob, off, lens), tasks in (w2, w3, w1):
        fresh = ctx.upload_reads(blob, off, lens, api.READS_NIBBLE)
        want = ctx.dynamic_programming(fresh, tasks)
        fresh.free()
        reads.refill(blob, off, lens)
        np.testing.assert_array_equal(ctx.dynamic_programming(reads, tasks), want)
        rc, orc = oracle.distance(packed, blob, off, lens, 0, tasks, -1)
        np.testing.assert_array_equal(want, orc)
    reads.free()
    st = torch.cuda.Stream()
    codes, packed, (blob, off, lens), tasks = w3
    r2 = ctx.upload_reads(blob, off, lens, api.READS_NIBBLE, stream=st.cuda_stream)
    rc, orc = oracle.distance(packed, blob, off, lens, 0, tasks, -1)
    np.testing.assert_array_equal(ctx.dynamic_programming(r2, tasks), orc)
    # the traceback's 64-bit vectors are rebuilt for the refilled reads
    codes, packed, (blob, off, lens), tasks = common.distance_workload(150, 13, 1500, seed=904, encoding=1, W=177)
    r3 = ctx.upload_reads(*w1[2], api.READS_NIBBLE)
    r3.free()
    r3 = ctx.upload_reads(blob, off, lens, 1)
    want = ctx.align(r3, tasks, 0, 1, cigar_stride=512)
    codes_b, packed_b, (blob_b, off_b, lens_b), tasks_b = common.distance_workload(150, 13, 1500, seed=905, encoding=1, W=177)
    r3.refill(blob_b, off_b, lens_b)
    ctx.align(r3, tasks_b, 0, 1, cigar_stride=512)
    r3.refill(blob, off, lens)
    got = ctx.align(r3, tasks, 0, 1, cigar_stride=512)
    np.testing.assert_array_equal(got[0], want[0])
    for i in np.nonzero(want[0]["ok"])[0]:
        e = want[0]["cigar_len"][i] + 1
        assert bytes(got[1][i, :e]) == bytes(want[1][i, :e]), i
    r2.free()
    r3.free()


def test_reads_refill_async(ctx, oracle):
    """wamd_reads_refill_async: stated sizes instead of a host pass, no wait.  Page-locked and pageable sources, a
    max_len above the longest read (more dwords per vector: same answers), mixed lengths, LevDiag and the alignment on
    the refilled batch; a read that breaks what was stated raises the malformed-task bit and cannot reach past its
    buffers."""
    import torch
    st = torch.cuda.Stream()
    w1 = common.distance_workload(150, 12, 3000, seed=911)
    w2 = common.distance_workload(100, 10, 5000, seed=912)
    w3 = common.distance_workload(250, 17, 700, seed=913)
    reads = ctx.upload_reads(*w1[2], api.READS_NIBBLE)
    for (codes, packed, (blob, off, lens), tasks), max_len in ((w2, 100), (w3, 256), (w1, 150), (w1, 190)):
        rc, want = oracle.distance(packed, blob, off, lens, 0, tasks, -1)
        pb, po, pl = api.PinnedBuffer(len(blob) + 64, np.uint8), api.PinnedBuffer(len(lens), np.uint64), \
            api.PinnedBuffer(len(lens), np.uint32)
        pb.array[:len(blob)] = blob
        po.array[:] = off[:len(lens)]
        pl.array[:] = lens
        reads.refill_async(pb.array, len(blob), po.array, pl.array, max_len, stream=st.cuda_stream)
        st.synchronize()
        assert ctx.take_flags(st.cuda_stream) == 0
        np.testing.assert_array_equal(ctx.dynamic_programming(reads, tasks), want)
        for b in (pb, po, pl):
            b.free()
    # pageable arrays (the copies are then synchronous, the call still correct), other strands, the alignment
    codes, packed, (blob, off, lens), tasks = common.distance_workload(150, 13, 1500, seed=914, encoding=1, W=177)
    r3 = ctx.upload_reads(*common.distance_workload(120, 9, 900, seed=915, encoding=1)[2], 1)
    want = oracle.align(packed, blob, off, lens, 1, tasks, 0, 1, cigar_stride=512)
    off_n = np.ascontiguousarray(off[:len(lens)])
    r3.refill_async(blob, len(blob), off_n, lens, 150, stream=st.cuda_stream)
    st.synchronize()
    got = ctx.align(r3, tasks, 0, 1, cigar_stride=512)
    np.testing.assert_array_equal(got[0], want[1])
    for i in np.nonzero(want[1]["ok"])[0]:
        e = want[1]["cigar_len"][i] + 1
        assert bytes(got[1][i, :e]) == bytes(want[2][i, :e]), i
    r3.free()
    # mixed lengths + LevDiag (direct orientation only)
    codes, packed, (blob, off, lens), tasks = common.levdiag_odd_workload(2000, 77)
    rc, want = oracle.levdiag(packed, codes.numel(), 0, blob, off, lens, tasks)
    r4 = ctx.upload_reads(*common.levdiag_workload(100, 3, 25, 50, 500, seed=5)[2], 0, orientations=1)
    off_n = np.ascontiguousarray(off[:len(lens)])
    r4.refill_async(blob, len(blob), off_n, lens, int(lens.max()), stream=st.cuda_stream)
    st.synchronize()
    np.testing.assert_array_equal(ctx.lev_diag(r4, tasks), want)
    r4.free()
    # reads that break the statements: longer than max_len, empty, reaching past the block
    codes, packed, (blob, off, lens), tasks = w1
    for what in ("long", "empty", "past"):
        l2, o2 = lens.copy(), np.ascontiguousarray(off[:len(lens)]).copy()
        if what == "long":
            l2[17] = 900
        elif what == "empty":
            l2[18] = 0
        else:
            o2[19] = len(blob) - 10
        reads.refill_async(blob, len(blob), o2, l2, 150, stream=st.cuda_stream)
        st.synchronize()
        assert ctx.take_flags(st.cuda_stream) & 32, what
        with pytest.raises(api.WamdError):
            reads.refill_async(blob, 0, o2, l2, 150, stream=st.cuda_stream)
    # and the batch is good again after a clean refill
    reads.refill_async(blob, len(blob), np.ascontiguousarray(off[:len(lens)]), lens, 150, stream=st.cuda_stream)
    st.synchronize()
    assert ctx.take_flags(st.cuda_stream) == 0
    rc, want = oracle.distance(packed, blob, off, lens, 0, tasks, -1)
    np.testing.assert_array_equal(ctx.dynamic_programming(reads, tasks), want)
    reads.free()
