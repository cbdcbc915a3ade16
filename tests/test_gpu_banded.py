"""Accepted-only mode of the distance path (banded kernels, WAMD_DISTANCE_ACCEPTED_ONLY) vs the
oracle: every candidate the reference accepts (edit_distance <= k) must come back with the
reference's distance AND end column (variant tie-break included); every other one as {k + 1, 0}.
Bit-exact bar.  The shapes walk across the band limits: 64-row band (n_cols - m + 2k + 2 <= 64),
96-row band, whole columns, and waves that mix lengths, rules and windows."""
import numpy as np
import pytest
import torch

import common
from whisper_amd import api, synth

pytestmark = pytest.mark.gpu

ACCEPTED_ONLY = 1


@pytest.fixture(scope="module")
def ctx():
    codes, packed = common.genome()
    c = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    c.set_distance_mode(ACCEPTED_ONLY)
    yield c
    c.destroy()


def clamp(want, k):
    k = np.asarray(k, np.uint32)
    out = want.copy()
    rej = want["edit_distance"] > k
    out["edit_distance"] = np.where(rej, k + 1, want["edit_distance"])
    out["pos"] = np.where(rej, 0, want["pos"])
    return out


def check(ctx, oracle, packed, blob, off, lens, enc, tasks, last, min_accepted):
    rc, want = oracle.distance(packed, blob, off, lens, enc, tasks, last)
    assert rc == 0
    want = clamp(want, tasks["max_ed"])
    reads = ctx.upload_reads(blob, off, lens, enc)
    got = ctx.dynamic_programming(reads, tasks, last)
    reads.free()
    assert (want["edit_distance"] <= tasks["max_ed"]).sum() >= min_accepted
    bad = np.flatnonzero((got["edit_distance"] != want["edit_distance"]) | (got["pos"] != want["pos"]))
    assert len(bad) == 0, "first differences (task, got, want): %s" % [
        (int(i), tuple(got[i]), tuple(want[i])) for i in bad[:5]]


BAND_CASES = [
    # m, k, W, variant, encoding, last_symbol
    (150, 12, None, 0, 0, -1),      # S1: 64-row band, 128 sliding columns
    (150, 12, None, 0, 1, 0),       # ASCII reads, stale last_symbol
    (150, 12, None, 0, 1, 3),
    (150, 5, None, 0, 0, -1),
    (150, 1, None, 0, 0, -1),
    (150, 0, None, 0, 0, -1),       # nothing can be accepted: the skewed loop stops before column m
    (150, 15, None, 0, 0, -1),      # last k of the 64-row band at W = m + 2k
    (150, 16, None, 0, 0, -1),      # 96-row band
    (150, 23, None, 0, 0, -1),      # last k of the 96-row band
    (150, 24, None, 0, 0, -1),      # whole columns
    (150, 13, 177, 0, 1, 0),        # fill_cigar_with_lev window
    (150, 12, 150, 0, 0, -1),       # window as long as the read
    (150, 12, 145, 0, 0, -1),       # shorter than the read
    (151, 12, None, 0, 0, -1),
    (250, 7, None, 0, 0, -1),       # S5: 64-row band on 8 dwords
    (250, 17, None, 0, 0, -1),      # S5 sensitive: 96-row band
    (100, 10, None, 0, 0, -1),      # M128 tie-break
    (100, 3, None, 0, 0, -1),
    (127, 8, None, 0, 0, -1),
    (128, 8, None, 0, 0, -1),
    (193, 8, None, 0, 0, -1),
    (255, 10, None, 0, 0, -1),
    (256, 10, None, 0, 0, -1),      # M64: W columns
    (300, 20, None, 0, 0, -1),
    (300, 12, None, 0, 0, -1),
    (150, 12, None, 1, 0, -1),      # M64 forced on a short read
    (150, 26, 660, 0, 1, 0),        # mate rescue window: nothing fits a band
    (64, 6, None, 0, 0, -1),        # short reads: whole columns
    (30, 3, None, 0, 0, -1),
    (97, 9, None, 0, 0, -1),
]


@pytest.mark.parametrize("m,k,W,variant,enc,last", BAND_CASES)
def test_accepted_only(ctx, oracle, m, k, W, variant, enc, last):
    codes, packed, (blob, off, lens), tasks = common.distance_workload(m, k, 6000, seed=500 + m + 7 * k,
                                                                      encoding=enc, W=W, variant=variant)
    check(ctx, oracle, packed, blob, off, lens, enc, tasks, last, 0 if (W or 1 << 30) <= m or k == 0 else 800)


@pytest.mark.parametrize("m,k,frac", [(150, 12, 0.12), (150, 12, 0.02), (150, 5, 0.1), (250, 17, 0.12), (250, 7, 0.05),
                                      (100, 10, 0.1), (150, 12, 0.97)])
def test_accepted_only_sparse_and_dense_batches(ctx, oracle, m, k, frac):
    """The mixes at which a block of the banded kernels gives waves back: few related candidates among unrelated loci
    (a mapper's batches: one, two or three of a block's four waves leave at the test column, the survivors change
    lanes), and nearly all related (nobody leaves, nothing moves)."""
    codes, packed, (blob, off, lens), tasks = common.distance_workload(m, k, 7000, seed=900 + m + k, true_frac=frac)
    check(ctx, oracle, packed, blob, off, lens, 0, tasks, -1, 20)


def test_accepted_only_mixed_waves(ctx, oracle):
    """Lengths, budgets and windows vary inside every wave: different tie-break rules, band
    fits and fixed-phase starts side by side (deferred lanes go down the chain)."""
    codes, packed = common.genome()
    g = torch.Generator().manual_seed(11)
    lens_ = [150, 151, 150, 149, 151, 150, 100, 250, 127, 128, 150, 256, 300, 33, 150, 151]
    reads_codes, tasks = [], []
    for i, m in enumerate(lens_ * 250):
        p = int(torch.randint(20000, 900000, (1,), generator=g))
        r = codes[p:p + m].clone()
        ne = int(torch.randint(0, 9, (1,), generator=g))
        for _ in range(ne):
            x = int(torch.randint(0, m, (1,), generator=g))
            kind = int(torch.randint(0, 10, (1,), generator=g))
            if kind < 7:
                r[x] = (int(r[x]) + 1 + int(torch.randint(0, 3, (1,), generator=g))) & 3
            elif kind == 7 and x + 1 < m:      # deletion from the read
                r[x:-1] = r[x + 1:].clone()
            elif x + 1 < m:                    # insertion into the read
                r[x + 1:] = r[x:-1].clone()
        r[r > 3] = 4
        reads_codes.append(r)
        k = int(torch.randint(0, 14, (1,), generator=g))
        k = min(k, m - 1)
        extra = int(torch.randint(0, 6, (1,), generator=g))
        tasks.append((i, p - k, m + 2 * k + extra, k, 0, 0))
    blob, off, lens = common.mixed_reads(reads_codes, 0)
    tasks = np.array(tasks, api.DIST_TASK)
    check(ctx, oracle, packed, blob, off, lens, 0, tasks, -1, 1000)


def test_accepted_only_n_runs(oracle):
    """Windows full of N_ref runs: the per-lane slow groups of both phases."""
    codes = synth.make_genome(1 << 18, seed=9, n_frac=0.05, n_run=7)
    packed = synth.pack_nibbles(codes).numpy()
    c = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    c.set_distance_mode(ACCEPTED_ONLY)
    try:
        for m, k, enc in ((150, 12, 0), (150, 12, 1), (250, 17, 0), (100, 10, 0), (300, 12, 0)):
            rd = synth.make_reads(codes, 4000, m, k, seed=40 + m, n_read_frac=0.2)
            blob, off, lens = common.host_reads(rd, enc)
            tasks = synth.distance_tasks(rd, m, k)
            # ASCII reads come through preprocessRawSeq, which leaves last_symbol alone (0 in CSamGenerator)
            check(c, oracle, packed, blob, off, lens, enc, tasks, 0 if enc else -1, 300)
    finally:
        c.destroy()


def test_accepted_only_low_complexity(oracle):
    """Tandem repeats: many co-optimal end columns, so the tie-break rules decide."""
    codes, packed = common.low_complexity_genome()
    c = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    c.set_distance_mode(ACCEPTED_ONLY)
    try:
        for m, k in ((150, 12), (100, 10), (250, 17), (300, 20), (150, 15)):
            rd = synth.make_reads(codes, 5000, m, k, seed=60 + m + k, true_frac=0.9)
            blob, off, lens = common.host_reads(rd, 0)
            tasks = synth.distance_tasks(rd, m, k)
            check(c, oracle, packed, blob, off, lens, 0, tasks, -1, 1500)
    finally:
        c.destroy()


def test_mode_switch_and_errors(ctx, oracle):
    codes, packed, (blob, off, lens), tasks = common.distance_workload(150, 12, 600, seed=3)
    reads = ctx.upload_reads(blob, off, lens, 0)
    bad = tasks.copy()
    bad["max_distance_in_ref"][5] = 0
    with pytest.raises(api.WamdError) as e:
        ctx.dynamic_programming(reads, bad)
    assert e.value.status == 5
    rc, want = oracle.distance(packed, blob, off, lens, 0, tasks, -1)
    ctx.set_distance_mode(0)
    try:
        np.testing.assert_array_equal(ctx.dynamic_programming(reads, tasks), want)
    finally:
        ctx.set_distance_mode(ACCEPTED_ONLY)
    np.testing.assert_array_equal(ctx.dynamic_programming(reads, tasks), clamp(want, tasks["max_ed"]))
    with pytest.raises(api.WamdError):
        ctx.set_distance_mode(7)
    reads.free()


def test_accepted_only_on_several_streams(ctx, oracle):
    """Launches on different streams of one context overlap: each stream has its own pair of
    deferral flags, so a batch that defers everything (rescue windows) next to one that defers
    nothing must not disturb it.  More streams than flag pairs fall back to the flag-less chain."""
    dev = torch.device("cuda", 0)
    shapes = [(150, 12, None), (150, 26, 660), (250, 17, None)]
    work = []
    for i, (m, k, W) in enumerate(shapes):
        codes, packed, (blob, off, lens), tasks = common.distance_workload(m, k, 20000, seed=900 + i, W=W)
        rc, want = oracle.distance(packed, blob, off, lens, 0, tasks, -1)
        assert rc == 0
        reads = ctx.upload_reads(blob, off, lens, 0)
        d_t = torch.from_numpy(tasks.view(np.uint8)).to(dev)
        work.append((reads, d_t, len(tasks), clamp(want, tasks["max_ed"])))
    streams = [torch.cuda.Stream(device=dev) for _ in range(20)]
    outs = []
    for rep in range(2):
        for si, st in enumerate(streams):
            reads, d_t, n, want = work[si % len(work)]
            outs.append((torch.full((n * 2,), -1, dtype=torch.int32, device=dev), want, st, work[si % len(work)]))
    torch.cuda.synchronize()   # the fills above run on torch's stream
    for d_r, want, st, (reads, d_t, n, _) in outs:
        ctx.dynamic_programming_dev(reads, d_t, n, d_r, last_symbol=-1, stream=st.cuda_stream)
    torch.cuda.synchronize()
    assert ctx.take_flags() == 0
    for d_r, want, _, _ in outs:
        got = d_r.cpu().numpy().view(np.uint32).reshape(-1, 2)
        np.testing.assert_array_equal(got[:, 0], want["edit_distance"])
        np.testing.assert_array_equal(got[:, 1], want["pos"])
    for reads, _, _, _ in work:
        reads.free()


def test_accepted_only_equals_exact_at_scale():
    """Size-independent property at 2^20 candidates per shape, generated on the GPU: the banded
    chain returns the exact kernels' answer for every accepted candidate and {k + 1, 0} for every
    other one (the exact kernels are pinned to the oracle by test_gpu_parity / test_gpu_golden)."""
    dev = torch.device("cuda", 0)
    n = 1 << 20
    codes = synth.make_genome(1 << 24, seed=77, device=dev)
    packed = synth.pack_nibbles(codes)
    c = api.Context(packed, ref_size=codes.numel(), cur_offset=0, device=0)
    try:
        for m, k, extra in ((150, 12, 0), (150, 15, 0), (151, 5, 3), (250, 17, 0), (250, 7, 1), (100, 10, 0),
                            (300, 20, 0), (150, 13, 1)):
            rd = synth.make_reads(codes, n, m, k, seed=1000 + m + k, chunk=1 << 15)
            blob, off, lens = synth.pack_reads(rd["codes"], 0)
            reads = c.upload_reads(blob.cpu().numpy(), off.cpu().numpy().astype(np.uint64),
                                   lens.cpu().numpy().astype(np.uint32), api.READS_NIBBLE, orientations=3)
            t = synth.distance_tasks(rd, m, k, W=m + 2 * k + extra)
            d_t = torch.from_numpy(t.view(np.uint8)).to(dev)
            out = []
            for mode in (0, ACCEPTED_ONLY):
                c.set_distance_mode(mode)
                d_r = torch.zeros(n * 2, dtype=torch.int32, device=dev)
                c.dynamic_programming_dev(reads, d_t, n, d_r, last_symbol=-1)
                c.synchronize()
                out.append(d_r.cpu().numpy().view(np.uint32).reshape(-1, 2))
            c.set_distance_mode(0)
            assert c.take_flags() == 0
            exact, banded = out
            rej = exact[:, 0] > k
            assert (~rej).sum() > n // 4
            bad = np.flatnonzero((banded[~rej] != exact[~rej]).any(axis=1))
            assert len(bad) == 0, "m=%d k=%d: accepted candidates differ, first %s" % (m, k, bad[:5])
            assert (banded[rej, 0] == k + 1).all() and (banded[rej, 1] == 0).all()
            reads.free()
    finally:
        c.destroy()


@pytest.mark.parametrize("m,k,W,refine", [(150, 12, None, 1), (150, 13, 177, 0), (250, 17, None, 1), (100, 10, None, 1),
                                          (150, 26, 660, 1), (300, 20, None, 1)])
def test_align_takes_its_first_pass_in_the_contexts_mode(ctx, oracle, m, k, W, refine):
    """getExtCigar only runs behind a dynamicProgramming that returned true (sam_matching.cpp:704-705): in
    accepted-only mode every accepted task's alignment is the oracle's, byte for byte; a rejected task
    comes back not ok with {k + 1, 0} and an empty string."""
    codes, packed, (blob, off, lens), tasks = common.distance_workload(m, k, 5000, seed=1700 + m + k, encoding=1, W=W,
                                                                      true_frac=0.6)
    stride = 4 * ((2 * (m + k) + 8 + 3) // 4)
    rc, want, wcig = oracle.align(packed, blob, off, lens, 1, tasks, 0, refine, cigar_stride=stride)
    assert rc == 0
    acc = want["ok"] == 1
    assert acc.sum() > 1000 and (~acc).sum() > 500
    reads = ctx.upload_reads(blob, off, lens, 1)
    got, gcig = ctx.align(reads, tasks, 0, refine, cigar_stride=stride)
    reads.free()
    for f in want.dtype.names:
        if f != "reserved":
            np.testing.assert_array_equal(got[f][acc], want[f][acc], err_msg=f)
    for i in np.flatnonzero(acc):
        n = want["cigar_len"][i]
        assert bytes(gcig[i, :n + 1]) == bytes(wcig[i, :n + 1]), i
    rej = ~acc
    assert (got["ok"][rej] == 0).all() and (got["edit_distance"][rej] == k + 1).all()
    assert (got["end_pos"][rej] == 0).all() and (got["cigar_len"][rej] == 0).all() and (gcig[rej, 0] == 0).all()


def test_finish_pairs_in_accepted_only_mode(ctx, oracle):
    """wamd_finish_pairs re-verifies lev records with k = err_edit_distance (fill_cigar_with_lev,
    sam_matching.cpp:689-752): the banded first pass gives the same SAM-visible output, and a record
    that claims fewer errors than its read has fails the same way."""
    from oracle import bindings as ob
    (codes, packed), (blob, off, lens), desc, ridx, maps, direct, distant, stale = common.finish_workload(
        oracle, 1500, 57, m=150, flank=6)
    tasks = common.finish_tasks(desc, ridx, lens, maps, direct, distant, stale)
    lev = np.nonzero((tasks["type"] == 0) & (tasks["err_edit_distance"] > 2))[0]
    tasks["err_edit_distance"][lev[:20]] = 0
    tasks["penalty"][lev[:20]] = 1
    rc, *want = oracle.finish_pairs(packed, blob, off, lens, tasks, ob.default_finish_params(6), ext_stride=768)
    assert rc == 0 and 0 < (want[0]["ok"] == 0).sum() <= 40
    c = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    try:
        c.set_distance_mode(ACCEPTED_ONLY)
        reads = c.upload_reads(blob, off, lens, 1, orientations=3)
        got = c.finish_pairs(reads, tasks, api.default_finish_params(6), ext_stride=768)
        common.same_finish_output(got, want)
        reads.free()
    finally:
        c.destroy()
