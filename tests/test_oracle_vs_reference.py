"""The CPU restatement against the reference's own objects on fresh seeded tasks.
Runs only where oracle/_ref/libwhisper_ref.so exists (built from /root/reference)."""
import numpy as np
import pytest

import common

SHAPES = [(150, 12, None, 0, 0, -1), (150, 12, None, 0, 1, 0), (150, 12, None, 0, 1, 2),
          (250, 17, None, 0, 0, -1), (100, 10, None, 0, 0, -1), (300, 20, None, 0, 0, -1),
          (64, 6, None, 0, 0, -1), (65, 6, None, 0, 0, -1), (127, 8, None, 0, 0, -1),
          (128, 8, None, 0, 0, -1), (255, 10, None, 0, 0, -1), (256, 10, None, 0, 0, -1),
          (150, 12, None, 1, 0, -1), (150, 26, 660, 0, 1, 0), (250, 44, 700, 0, 1, 0),
          (250, 7, None, 0, 0, -1), (150, 26, 1427, 0, 1, 0)]


@pytest.mark.parametrize("m,k,W,variant,enc,last", SHAPES)
def test_distance(oracle, reference, m, k, W, variant, enc, last):
    codes, packed, (blob, off, lens), tasks = common.distance_workload(m, k, 8000, seed=11 * m + k,
                                                                      encoding=enc, W=W, variant=variant)
    rc1, a = oracle.distance(packed, blob, off, lens, enc, tasks, last)
    rc2, b = reference.distance(packed, codes.numel(), blob, off, lens, enc, tasks, last)
    assert rc1 == 0 and rc2 == 0
    np.testing.assert_array_equal(a, b)


def test_distance_text_len_error(oracle, reference):
    codes, packed, (blob, off, lens), tasks = common.distance_workload(150, 12, 10, seed=5)
    tasks["max_distance_in_ref"][3] = 0
    rc1, _ = oracle.distance(packed, blob, off, lens, 0, tasks, -1)
    rc2, _ = reference.distance(packed, codes.numel(), blob, off, lens, 0, tasks, -1)
    assert rc1 == 5 and rc2 == 5      # WAMD_ERR_TEXT_LEN: the reference threw runtime_error


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6])
def test_levdiag(oracle, reference, k):
    for j, (fl, fr) in enumerate(common.stage_segments(150, k)):
        codes, packed, (blob, off, lens), tasks = common.levdiag_workload(150, k, fl, fr, 4000,
                                                                         seed=500 + 10 * k + j)
        _, a = oracle.levdiag(packed, codes.numel(), 0, blob, off, lens, tasks)
        _, b = reference.levdiag(packed, codes.numel(), 0, blob, off, lens, tasks)
        np.testing.assert_array_equal(a, b)


def test_levdiag_odd_shapes(oracle, reference):
    """Short and empty anchors, missing sides, k = 0, N runs facing reads with equal and unequal N codes."""
    codes, packed, (blob, off, lens), tasks = common.levdiag_odd_workload(6000, 4242)
    _, a = oracle.levdiag(packed, codes.numel(), 0, blob, off, lens, tasks)
    _, b = reference.levdiag(packed, codes.numel(), 0, blob, off, lens, tasks)
    assert (a <= tasks["max_mismatches"]).sum() > 1500
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("m,ks", [(250, (1, 2, 3, 4)), (300, (4, 6)), (600, (3, 6)), (1000, (6,))])
def test_levdiag_long_reads(oracle, reference, m, ks):
    """The shapes of the GPU test of the same name (reads beyond 160 symbols; 250 bp = BASELINE configs[4])."""
    for k in ks:
        for j, (fl, fr) in enumerate(common.stage_segments(m, k)):
            codes, packed, (blob, off, lens), tasks = common.levdiag_workload(m, k, fl, fr, 1500,
                                                                             seed=1200 + m + 10 * k + j)
            _, a = oracle.levdiag(packed, codes.numel(), 0, blob, off, lens, tasks)
            _, b = reference.levdiag(packed, codes.numel(), 0, blob, off, lens, tasks)
            assert (a <= k).sum() > 150
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("k", [2, 4, 6])
def test_levdiag_low_complexity(oracle, reference, k):
    """The same tandem-repeat workload the GPU test uses: pins the oracle where alignments are ambiguous."""
    gen = common.low_complexity_genome()
    codes, packed = gen
    for j, (fl, fr) in enumerate(common.stage_segments(120, k)):
        _, _, (blob, off, lens), tasks = common.levdiag_workload(120, k, fl, fr, 4000, seed=900 + 10 * k + j, gen=gen)
        _, a = oracle.levdiag(packed, codes.numel(), 0, blob, off, lens, tasks)
        _, b = reference.levdiag(packed, codes.numel(), 0, blob, off, lens, tasks)
        assert (a <= k).sum() > 300
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("m,k,W,last,refine", [(150, 12, None, 0, 1), (150, 13, 177, 0, 1),
                                               (150, 26, 660, 0, 1), (250, 17, None, 0, 1),
                                               (100, 10, None, 0, 0), (300, 20, None, 0, 1),
                                               (150, 12, None, 3, 1), (150, 26, 1427, 0, 1),
                                               (250, 44, 1588, 0, 1)])
def test_align(oracle, reference, m, k, W, last, refine):
    codes, packed, (blob, off, lens), tasks = common.distance_workload(m, k, 4000, seed=7 * m + k,
                                                                      encoding=1, W=W)
    _, a, ca = oracle.align(packed, blob, off, lens, 1, tasks, last, refine)
    _, b, cb = reference.align(packed, codes.numel(), blob, off, lens, 1, tasks, last, refine)
    for f in a.dtype.names:
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)
    assert (ca == cb).all()


def test_align_planted_refinement(oracle, reference):
    codes, packed, (blob, off, lens), tasks = common.planted_refine_workload(600, 150, 12, seed=99)
    _, a, ca = oracle.align(packed, blob, off, lens, 1, tasks, 0, 1)
    _, b, cb = reference.align(packed, codes.numel(), blob, off, lens, 1, tasks, 0, 1)
    assert b["refined"].sum() > 100
    for f in a.dtype.names:
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)
    assert (ca == cb).all()


@pytest.mark.parametrize("enc,mcf", [(0, 1.0), (0, 0.6), (1, 1.0), (1, 0.5)])
def test_indel_match_cigar_serialize(oracle, reference, enc, mcf):
    from oracle import bindings
    codes, packed, (blob, off, lens), tasks = common.indel_workload(2500, 150, seed=31 + enc, encoding=enc)
    prm = bindings.default_indel_params(mcf)
    _, a = oracle.indel_match(packed, blob, off, lens, enc, tasks, prm)
    _, b = reference.indel_match(packed, blob, off, lens, enc, tasks, prm)
    assert b["found"].sum() > 800
    np.testing.assert_array_equal(a, b)
    ct = common.indel_cigar_tasks(tasks, b)
    _, ra, ca = oracle.indel_ext_cigar(packed, blob, off, lens, enc, ct)
    _, rb, cb = reference.indel_ext_cigar(packed, blob, off, lens, enc, ct)
    np.testing.assert_array_equal(ra, rb)
    np.testing.assert_array_equal(ca, cb)
    d = (np.arange(len(b)) % 2).astype(np.uint8)
    _, wa, sa = oracle.serialize(b, d)
    _, wb, sb = reference.serialize(b, d)
    np.testing.assert_array_equal(wa, wb)
    np.testing.assert_array_equal(sa, sb)


def test_indel_match_m100_and_m250(oracle, reference):
    for m in (100, 250):
        codes, packed, (blob, off, lens), tasks = common.indel_workload(1200, m, seed=m, stage_major=4)
        _, a = oracle.indel_match(packed, blob, off, lens, 0, tasks)
        _, b = reference.indel_match(packed, blob, off, lens, 0, tasks)
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("m,enc,mcf,stage,max_indel", [(300, 0, 0.6, 8, 50), (150, 0, 0.6, 5, 100), (150, 0, 0.6, 5, 80),
                                                        (600, 1, 0.7, 8, 120)])
def test_indel_match_long_reads_and_wide_indels(oracle, reference, m, enc, mcf, stage, max_indel):
    """The shapes that run indel_match_kernel<false> on the GPU (reads above 256, max_indel_size above 80)."""
    from oracle import bindings
    codes, packed, (blob, off, lens), tasks = common.indel_workload(2000, m, seed=400 + m + enc + max_indel, encoding=enc,
                                                                   stage_major=stage, max_indel=max_indel,
                                                                   max_mm=12 if m <= 250 else 20)
    prm = bindings.default_indel_params(mcf)
    _, a = oracle.indel_match(packed, blob, off, lens, enc, tasks, prm)
    _, b = reference.indel_match(packed, blob, off, lens, enc, tasks, prm)
    assert b["found"].sum() > 500
    np.testing.assert_array_equal(a, b)
    ct = common.indel_cigar_tasks(tasks, b)
    _, ra, ca = oracle.indel_ext_cigar(packed, blob, off, lens, enc, ct, cigar_stride=2048)
    _, rb, cb = reference.indel_ext_cigar(packed, blob, off, lens, enc, ct, max_indel=max_indel, cigar_stride=2048)
    np.testing.assert_array_equal(ra, rb)
    np.testing.assert_array_equal(ca, cb)


def test_clip_boundary(oracle, reference):
    rec, cig = common.clip_workload(2500, 150, seed=17)
    _, ra, ca = oracle.clip_boundary(rec, cig)
    _, rb, cb = reference.clip_boundary(rec, cig)
    assert (cb != cig).any(axis=1).sum() > 1000
    for f in ra.dtype.names:
        if f != "no_clippings":
            np.testing.assert_array_equal(ra[f], rb[f], err_msg=f)
    np.testing.assert_array_equal(ca, cb)


@pytest.mark.parametrize("k,cur_offset,min_len", [(5, 0, 150), (2, 0, 150), (5, 31, 150), (7, 0, 250), (3, 62, 250),
                                                   (4, 9, 100), (8, 33, 300)])
def test_cf_filter(oracle, reference, k, cur_offset, min_len):
    codes, packed, (blob, off, lens), tasks = common.cf_workload(3000, seed=40 + k, k=k, cur_offset=cur_offset,
                                                                 min_len=min_len)
    rc1, a = oracle.cf_filter(packed, codes.numel(), cur_offset, blob, off, lens, tasks)
    rc2, b, ok = reference.cf_filter(packed, codes.numel(), cur_offset, blob, off, lens, tasks, max_diff=4 * k)
    assert rc1 == 0 and rc2 == 0
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ok, (a <= 4 * k).astype(np.uint8))
    assert 300 < ok.sum() < 2900     # the filter both passes and rejects


def test_sam_cigar(oracle, reference):
    """Row X: extract_cigar_and_mdz / extract_cigarbin_and_mdz (sam.cpp:805-1051)."""
    ext = common.ext_cigar_pool(oracle)
    rc1, c1, m1, b1, l1 = oracle.sam_cigar(ext)
    rc2, c2, m2, b2, l2 = reference.sam_cigar(ext)
    assert rc1 == 0 and rc2 == 0
    np.testing.assert_array_equal(l1, l2)
    for i in range(len(ext)):
        assert bytes(c1[i, :l1["cigar_len"][i] + 1]) == bytes(c2[i, :l2["cigar_len"][i] + 1])
        assert bytes(m1[i, :l1["mdz_len"][i] + 1]) == bytes(m2[i, :l2["mdz_len"][i] + 1])
        nb = l1["cigar_bin_len"][i]
        np.testing.assert_array_equal(b1[i, :nb], b2[i, :nb])
    assert (l1["cigar_bin_len"] > 3).sum() > 200       # plenty of multi-operation CIGARs


def _same_clip_output(ra, ca, rb, cb, skip=()):
    for f in ra.dtype.names:
        if f not in ("reserved",) + tuple(skip):
            np.testing.assert_array_equal(ra[f], rb[f], err_msg=f)
    for i in range(len(ra)):          # as C strings: the reference leaves stale bytes behind the NUL
        n = ra["cigar_len"][i] + 1
        assert bytes(ca[i, :n]) == bytes(cb[i, :n]), i


def test_clip_illegal(oracle, reference):
    """Row L: clipIllegalPositions (soft_clipping.cpp:592-729)."""
    t, rec, cig = common.clip_illegal_workload(oracle, 3000, seed=5)
    rc1, ra, ca = oracle.clip_illegal(t, rec, cig)
    rc2, rb, cb = reference.clip_illegal(t, rec, cig)
    assert rc1 == 0 and rc2 == 0
    assert (ra["no_clippings"] == 1).sum() > 800 and (ra["no_clippings"] == 2).sum() > 400
    assert (ra["no_clippings"] == 0).sum() > 400
    _same_clip_output(ra, ca, rb, cb)


def test_clip_overlapping_pairs(oracle, reference):
    """Row L: clipOverlappingPairs, clipLeft, clipRight (soft_clipping.cpp:997-1151)."""
    t, rec, cig = common.clip_pairs_workload(oracle, 3000, seed=7)
    rc1, ra, ca = oracle.clip_overlapping_pairs(t, rec, cig)
    rc2, rb, cb = reference.clip_overlapping_pairs(t, rec, cig)
    assert rc1 == 0 and rc2 == 0
    assert (ra["no_clippings"] > 0).sum() > 1500
    changed = (ra["left_clipping"] != rec["left_clipping"]) | (ra["right_clipping"] != rec["right_clipping"])
    assert changed.sum() > 1000
    _same_clip_output(ra, ca, rb, cb, skip=("no_clippings",))   # the reference does not say whether it clipped


@pytest.mark.parametrize("lowcx,fresh,enc,W", [(0, 0, 1, 700), (0, 1, 1, 1427), (1, 0, 1, 700), (1, 1, 0, 260),
                                                (1, 0, 0, 260)])
def test_softclip_match(oracle, reference, lowcx, fresh, enc, W):
    """Row S: CSoftClipping::{preprocess*, match, getExtCigar} (soft_clipping.cpp:139-589) - random genome
    and tandem repeats (shared probe chains, fake hits), fresh and long-lived object state."""
    from oracle import bindings as ob
    gen = common.low_complexity_genome() if lowcx else None
    codes, packed, (blob, off, lens), tasks = common.softclip_workload(1500, seed=10 + W + fresh, W=W, encoding=enc,
                                                                       gen=gen, lens_jitter=bool(fresh))
    prm = ob.default_softclip_params(fresh_object=fresh)
    rc1, a, ca = oracle.softclip_match(packed, blob, off, lens, enc, tasks, prm)
    rc2, b, cb = reference.softclip_match(packed, blob, off, lens, enc, tasks, prm)
    assert rc1 == 0 and rc2 == 0
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ca, cb)
    assert a["found"].sum() > 1000 and (a["del_size"] > 0).sum() > 30 and (a["del_size"] < 0).sum() > 30
    assert (a["no_mismatches"] > 0).sum() > 20 and (a["found"] == 0).sum() > 100


@pytest.mark.parametrize("m,W,mql", [(700, 2000, 1024), (1000, 3000, 1024)])
def test_softclip_match_long_reads(oracle, reference, m, W, mql):
    """The same for reads of 700 and 1 000 symbols (max_query_len 1 024: a table of 65 536 cells), the lengths the
    GPU path takes since round 3 - the oracle's table follows the reference's there as well."""
    from oracle import bindings as ob
    for gen in (None, common.low_complexity_genome()):
        codes, packed, (blob, off, lens), tasks = common.softclip_workload(200, seed=m + W, m=m, W=W, encoding=1, gen=gen,
                                                                           lens_jitter=True)
        prm = ob.default_softclip_params(max_query_len=mql, fresh_object=0)
        rc1, a, ca = oracle.softclip_match(packed, blob, off, lens, 1, tasks, prm, cigar_stride=1536)
        rc2, b, cb = reference.softclip_match(packed, blob, off, lens, 1, tasks, prm, cigar_stride=1536)
        assert rc1 == 0 and rc2 == 0
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(ca, cb)
        assert a["found"].sum() > 100


@pytest.mark.parametrize("seed,m", [(21, 150), (22, 250), (23, 100)])
def test_finish_pairs_chain(oracle, reference, seed, m):
    """The chosen pair from its wire records to the SAM fields (sam.cpp:1555-1580, 1871-1900, 1992-2007): the
    reference's own order on its own objects against the oracle's chain, and the Python construction of the
    descriptors (what the GPU tests feed the entry point) against the reference's deserialize + Translate +
    mapping_desc_t constructor."""
    (codes, packed), (blob, off, lens), desc, ridx, maps, direct, distant, stale = common.finish_workload(
        oracle, 500, seed, m=m)
    _, wire, _ = oracle.serialize(maps, direct)
    rc, tasks, res, ext, cig, mdz = reference.finish_pairs(packed, codes.numel(), desc, blob, off, lens, ridx, wire,
                                                           distant, stale, ext_stride=768)
    assert rc == 0 and res["ok"].all()
    mine = common.finish_tasks(desc, ridx, lens, maps, direct, distant, stale)
    for f in tasks.dtype.names:
        np.testing.assert_array_equal(mine[f], tasks[f], err_msg=f)
    rc, *got = oracle.finish_pairs(packed, blob, off, lens, tasks, ext_stride=768)
    assert rc == 0
    common.same_finish_output(got, (res, ext, cig, mdz))
    # the workload reaches every branch of the chain
    meth = np.bincount(res["method"], minlength=27)
    assert meth[0] + meth[1] > 100 and meth[2] + meth[3] > 400 and meth[13:26].sum() > 20
    assert (res["left_clipping"] > 0).sum() > 50 and (res["right_clipping"] > 0).sum() > 50
    assert (res["err_edit_distance"] != tasks["err_edit_distance"]).sum() > 50
    perfect_clipped = (res["method"] < 2) & ((res["left_clipping"] > 0) | (res["right_clipping"] > 0))
    assert perfect_clipped.sum() > 5            # a perfect match that a clipping pass gave an ext-CIGAR keeps "<len>M"


@pytest.mark.parametrize("seed,m", [(27, 150), (28, 100)])
def test_finish_pairs_distant_from_pair_and_failing_reads(oracle, reference, seed, m):
    """isDistant computed inside the chain (sam.cpp:1881-1884: for the second read from the first one's position and
    length after its fill_cigar and clipping passes), and reads fill_cigar_with_lev cannot align: the mate is still
    finished, the pair is not pair-clipped."""
    (codes, packed), (blob, off, lens), desc, ridx, maps, direct, distant, stale = common.finish_workload(
        oracle, 600, seed, m=m, from_pair_frac=0.6)
    lev = np.nonzero((maps["type"] == 0) & (maps["no_mismatches"] > 2))[0]
    maps["no_mismatches"][lev[::9]] = 1               # claims fewer errors than the read has: dynamicProgramming fails
    _, wire, _ = oracle.serialize(maps, direct)
    rc, tasks, res, ext, cig, mdz = reference.finish_pairs(packed, codes.numel(), desc, blob, off, lens, ridx, wire,
                                                           distant, stale, ext_stride=768, model=common.FINISH_MODEL)
    assert rc == 0
    bad = res["ok"] == 0
    assert 10 < bad.sum() < 200 and res["ok"][np.nonzero(bad)[0] ^ 1].sum() > 5      # mates of failing reads are finished
    mine = common.finish_tasks(desc, ridx, lens, maps, direct, distant, stale, isize=common.FINISH_ISIZE)
    for f in tasks.dtype.names:
        np.testing.assert_array_equal(mine[f], tasks[f], err_msg=f)
    rc, *got = oracle.finish_pairs(packed, blob, off, lens, tasks, ext_stride=768)
    assert rc == 0
    common.same_finish_output(got, (res, ext, cig, mdz))
    fp = (tasks["distant"] == 2) & (res["ok"] == 1)
    dist = np.isin(res["method"], (1, 3))
    assert (fp & dist).sum() > 50 and (fp & ~dist & np.isin(res["method"], (0, 2))).sum() > 50
    # the second read's answer follows the first one's alignment: some pairs differ between their two slots
    assert (dist[0::2] != dist[1::2])[fp[0::2] & fp[1::2]].sum() > 0


@pytest.mark.parametrize("seed,max_no_mappings,id_bytes", [(3, 1024, 5), (4, 8, 4), (5, 1, 8), (6, 3, 6)])
def test_commit_and_result_groups(oracle, reference, seed, max_no_mappings, id_bytes):
    """Verified candidates -> candidate_mapping (construct<lev>, better) -> the heap gatherer's order -> the bytes
    CMappingResultsCollector::Push leaves in a group buffer; the reference's own objects against the restatement."""
    from oracle import bindings as ob
    t, indel, read_first, read_ids = common.commit_workload(oracle, 3000, seed)
    rc1, a = oracle.commit(t, indel)
    rc2, b = reference.commit(t, indel)
    assert rc1 == 0 and rc2 == 0
    np.testing.assert_array_equal(a, b)
    assert (a["type"] == 0).sum() > 3000 and ((a["type"] > 0) & (a["type"] < 8)).sum() > 1000 and (a["found"] == 0).sum() > 1000
    prm = ob.GroupParams(id_bytes, 4, max_no_mappings, 0)
    rc1, ga = oracle.result_groups(t, a, read_first, read_ids, prm)
    rc2, gb = reference.result_groups(t, a, read_first, read_ids, prm)
    assert rc1 == 0 and rc2 == 0 and len(ga) > 20000
    assert bytes(ga) == bytes(gb)


def sa_loop_expected(oracle, packed, ref_size, reads2, n, tasks, k, start_pos, lev_alg, sensitive, prm=None):
    """What the loop body leaves per suffix-array entry, composed from the pinned oracle functions exactly as
    mapping_core_sa.cpp:223-344 composes them: CF filter (unless the read is short), LevDiag below lev_alg_thr = 7 else
    dynamicProgramming, Match in sensitive mode, construct<lev> / better / the penalty test."""
    cf, dist, lev, indel, commit, read_first = tasks
    blob, off, lens = reads2
    rc, val = oracle.cf_filter(packed, ref_size, start_pos, blob, off, lens, cf)
    assert rc == 0
    passed = (cf["pattern_len"] == 0) | (val <= 4 * k)
    if lev_alg:
        rc, ed = oracle.levdiag(packed, ref_size, start_pos, blob, off, lens, lev)
    else:
        rc, res = oracle.distance(packed, blob, off, lens, 0, dist, -1)
        ed = res["edit_distance"]
    assert rc == 0
    commit = commit.copy()
    commit["lev_errors"] = np.where(passed, ed, k + 1)
    im = None
    if sensitive:
        commit["has_indel"] = 1
        rc, im = oracle.indel_match(packed, blob, off, lens, 0, indel, prm)
        assert rc == 0
    rc, committed = oracle.commit(commit, im)
    assert rc == 0
    return commit, committed, passed


SA_LOOP_CASES = [
    # seed, m, k, stage segment index, sensitive (k is then stage_major * 2.5: M256 + CIndelMatching::Match)
    (1, 150, 5, 2, 0), (2, 150, 2, 0, 0), (3, 150, 12, 3, 1), (4, 100, 10, 1, 1), (5, 150, 5, 5, 0), (6, 250, 17, 4, 1),
]


@pytest.mark.parametrize("seed,m,k,segi,sensitive", SA_LOOP_CASES)
def test_sa_loops_against_the_composed_oracle(oracle, reference, seed, m, k, segi, sensitive):
    """CMappingCore::search_in_unknown_range_dir_and_rc / search_in_known_range_dir_and_rc themselves, on a real
    suffix array of a genome with near-repeats: every candidate_mapping_t they push to dir / rc_match_positions, in
    order, and the best penalty they return, against the composition the device entry points implement."""
    from oracle import bindings as ob
    stage = int(k / 2.5) if sensitive else k
    seg = common.stage_segments(m, stage)[segi]
    codes, packed, reads, sa_dir, sa_rc = common.sa_loop_workload(seed, 220, 3, m, k, seg, min_read_len=m)
    ref_size = codes.numel()
    prm = ob.default_indel_params(0.6 if seed & 1 else 1.0)
    rc, ranges, best, pushed, pdir, pread = reference.search_reads(packed, ref_size, *reads, sa_dir, sa_rc, seg, m, k,
                                                                  sensitive=sensitive, enable_indels=sensitive, params=prm)
    assert rc == 0
    n = len(reads[2])
    span = (ranges[:, 1] - ranges[:, 0]).astype(np.int64) + (ranges[:, 3] - ranges[:, 2])
    assert (span > 1).sum() > n // 10 and (span == 0).sum() < n // 3          # repeats: several entries per anchor
    reads2 = common.with_reverse_reads(reads)
    tasks = common.sa_loop_tasks(ranges, reads[2], sa_dir, sa_rc, ref_size, seg, m, k, 50)
    commit, committed, passed = sa_loop_expected(oracle, packed, ref_size, reads2, n, tasks, k, seg[0], k < 7, sensitive, prm)
    kept = committed["found"] == 1
    assert kept.sum() > n // 2 and (~kept).sum() > 50 and (~passed).sum() > 10
    # the loops' pushes = the kept candidates, in task order
    want = committed[kept]
    np.testing.assert_array_equal(pdir, commit["direct"][kept])
    read_of = np.repeat(np.arange(n), np.diff(tasks[5]).astype(np.int64))
    np.testing.assert_array_equal(pread, read_of[kept])
    for f in ("type", "pos", "no_mismatches", "matching_len1", "matching_len2", "indel_len1", "indel_len2", "clipping_len",
              "penalty"):
        np.testing.assert_array_equal(pushed[f], want[f], err_msg=f)
    if sensitive:
        assert (want["type"] != 0).sum() > 5
    # best_match of either strand: the smallest penalty among its candidates, kept or not (mapping_core_sa.cpp:334-335)
    for r in range(n):
        lo, hi = int(tasks[5][r]), int(tasks[5][r + 1])
        for strand, d in ((0, 1), (1, 0)):
            sel = np.arange(lo, hi)[commit["direct"][lo:hi] == d]
            pen = [int(committed["penalty"][i]) for i in sel if committed["found"][i]]
            assert int(best[r, strand]) == min(pen + [k + 1]), (r, strand)
