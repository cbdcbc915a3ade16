"""Non-default scoring_t through wamd_ctx_set_scoring / wamd_indel_params.scoring: the HIP path against the oracle
(pinned to the reference's own objects under the same scorings by tests/test_scoring_oracle.py and the *_sc_* fixtures).
One asymmetric scoring (insertions and deletions cost differently: SURVEY appendix A hazards 8, 9) and one whose
values are not multiples of 0.2 (hazard 11: sums are inexact, so the order of the double additions shows).  Scores
are compared for equality, as everywhere."""
import numpy as np
import pytest

import common
from whisper_amd import api

pytestmark = pytest.mark.gpu

NAMES = ["asym", "odd"]


@pytest.fixture(scope="module")
def ctx():
    codes, packed = common.genome()
    c = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    yield c
    c.destroy()


@pytest.fixture(params=NAMES)
def scored(request, ctx):
    """(context with the scoring set, oracle's Scoring, api's Scoring); the default is restored afterwards."""
    ctx.set_scoring(common.scoring(request.param, api))
    yield ctx, common.scoring(request.param), common.scoring(request.param, api)
    ctx.set_scoring(api.default_scoring())


@pytest.mark.parametrize("m,k,W,refine", [(150, 13, 177, 1), (150, 26, 660, 1), (250, 44, 700, 1), (100, 10, None, 0),
                                          (300, 20, None, 1)])
def test_align(scored, oracle, m, k, W, refine):
    ctx, sc, _ = scored
    codes, packed, (blob, off, lens), tasks = common.distance_workload(m, k, 3000, seed=500 + m + k, encoding=1, W=W)
    stride = 4 * ((2 * (m + k) + 8 + 3) // 4)
    rc, want, wcig = oracle.align(packed, blob, off, lens, 1, tasks, 0, refine, scoring=sc, cigar_stride=stride)
    assert rc == 0 and want["ok"].sum() > 400
    reads = ctx.upload_reads(blob, off, lens, 1)
    got, gcig = ctx.align(reads, tasks, 0, refine, cigar_stride=stride)
    reads.free()
    for f in want.dtype.names:
        np.testing.assert_array_equal(got[f], want[f], err_msg=f)
    for i in np.nonzero(want["ok"])[0]:
        n = want["cigar_len"][i]
        assert bytes(gcig[i, :n + 1]) == bytes(wcig[i, :n + 1]), i


def test_align_planted_refinement(scored, oracle):
    ctx, sc, _ = scored
    codes, packed, (blob, off, lens), tasks = common.planted_refine_workload(800, 150, 12, seed=97)
    rc, want, wcig = oracle.align(packed, blob, off, lens, 1, tasks, 0, 1, scoring=sc, cigar_stride=512)
    assert want["refined"].sum() > 150
    reads = ctx.upload_reads(blob, off, lens, 1)
    got, gcig = ctx.align(reads, tasks, 0, 1, cigar_stride=512)
    reads.free()
    for f in want.dtype.names:
        np.testing.assert_array_equal(got[f], want[f], err_msg=f)
    for i in np.nonzero(want["ok"])[0]:
        n = want["cigar_len"][i]
        assert bytes(gcig[i, :n + 1]) == bytes(wcig[i, :n + 1]), i


@pytest.mark.parametrize("m,enc,mcf,stage,max_indel", [(150, 0, 1.0, 5, 50), (150, 1, 0.6, 5, 50), (300, 0, 0.6, 8, 50),
                                                        (150, 0, 0.6, 5, 100)])
def test_indel_match_and_ext_cigar(scored, oracle, m, enc, mcf, stage, max_indel):
    """Match's costs are doubles compared with strict > in iteration order (indels.cpp:222-457); GetExtCigarNew's
    insertion score multiplies gap_ins_extend by a negative length (indels.cpp:728)."""
    from oracle import bindings
    ctx, sc, sc_api = scored
    codes, packed, (blob, off, lens), tasks = common.indel_workload(2000, m, seed=800 + m + enc + max_indel, encoding=enc,
                                                                   stage_major=stage, max_indel=max_indel,
                                                                   max_mm=12 if m <= 250 else 20)
    prm = bindings.default_indel_params(mcf)
    prm.scoring = sc
    _, want = oracle.indel_match(packed, blob, off, lens, enc, tasks, prm)
    reads = ctx.upload_reads(blob, off, lens, enc, orientations=1)
    got = ctx.indel_match(reads, tasks, api.default_indel_params(mcf, sc_api))
    assert want["found"].sum() > 500
    np.testing.assert_array_equal(got, want)
    ct = common.indel_cigar_tasks(tasks, want)
    stride = 4 * ((2 * (m + 2 * max_indel + 20) + 8 + 3) // 4)
    _, wres, wcig = oracle.indel_ext_cigar(packed, blob, off, lens, enc, ct, scoring=sc, cigar_stride=stride)
    gres, gcig = ctx.indel_ext_cigar(reads, ct, cigar_stride=stride)
    reads.free()
    np.testing.assert_array_equal(gres, wres)
    for i in range(len(wres)):
        n = wres["cigar_len"][i] + 1
        assert bytes(gcig[i, :n]) == bytes(wcig[i, :n]), i


def _same_clip_output(grec, gcig, wrec, wcig):
    for f in wrec.dtype.names:
        if f != "reserved":
            np.testing.assert_array_equal(grec[f], wrec[f], err_msg=f)
    for i in range(len(wrec)):
        n = wrec["cigar_len"][i] + 1
        assert bytes(gcig[i, :n]) == bytes(wcig[i, :n]), i


def test_clip_boundary_illegal_pairs(scored, oracle):
    ctx, sc, _ = scored
    rec, cig = common.clip_workload(3000, 150, seed=22, scoring=sc)
    _, wrec, wcig = oracle.clip_boundary(rec, cig, sc)
    grec, gcig = ctx.clip_boundary(rec, cig)
    assert (wrec["no_clippings"] > 0).sum() > 800
    _same_clip_output(grec, gcig, wrec, wcig)
    t, rec, cig = common.clip_illegal_workload(oracle, 3000, seed=35, scoring=sc)
    rc, wrec, wcig = oracle.clip_illegal(t, rec, cig, sc)
    assert rc == 0 and (wrec["no_clippings"] > 0).sum() > 1500
    grec, gcig = ctx.clip_illegal(t, rec, cig)
    _same_clip_output(grec, gcig, wrec, wcig)
    t, rec, cig = common.clip_pairs_workload(oracle, 2000, seed=36, scoring=sc)
    rc, wrec, wcig = oracle.clip_overlapping_pairs(t, rec, cig)
    assert rc == 0
    grec, gcig = ctx.clip_overlapping_pairs(t, rec, cig)
    _same_clip_output(grec, gcig, wrec, wcig)


def test_softclip_match(scored, oracle):
    from oracle import bindings as ob
    ctx, sc, _ = scored
    codes, packed, (blob, off, lens), tasks = common.softclip_workload(2000, seed=715, W=700, encoding=1)
    _, want, wcig = oracle.softclip_match(packed, blob, off, lens, 1, tasks, ob.default_softclip_params(fresh_object=0), sc)
    reads = ctx.upload_reads(blob, off, lens, 1, orientations=3)
    got, gcig = ctx.softclip_match(reads, tasks, api.default_softclip_params(fresh_object=0))
    reads.free()
    assert want["found"].sum() > 1200
    np.testing.assert_array_equal(got, want)
    for i in range(len(want)):
        n = want["cigar_len"][i] + 1
        assert bytes(gcig[i, :n]) == bytes(wcig[i, :n]), i


@pytest.mark.parametrize("seed,m,flank", [(35, 150, 6), (36, 100, 5)])
def test_finish_pairs_chain(scored, oracle, seed, m, flank):
    """The whole chain under the scoring: the mapping_desc_t constructor's indel_to_mismatch_score decides
    err_edit_distance (hence the k of fill_cigar_with_lev and whether boundary clipping runs at all), every pass adds
    its own doubles; AS:i: prints one decimal of the result (utils.h:276-286)."""
    from oracle import bindings as ob
    ctx, sc, _ = scored
    (codes, packed), (blob, off, lens), desc, ridx, maps, direct, distant, stale = common.finish_workload(
        oracle, 1200, seed, m=m, flank=flank, scoring=sc, from_pair_frac=0.3)
    tasks = common.finish_tasks(desc, ridx, lens, maps, direct, distant, stale, scoring=sc, isize=common.FINISH_ISIZE)
    rc, *want = oracle.finish_pairs(packed, blob, off, lens, tasks, ob.default_finish_params(flank), sc, ext_stride=768)
    assert rc == 0 and want[0]["ok"].all()
    reads = ctx.upload_reads(blob, off, lens, 1, orientations=3)
    got = ctx.finish_pairs(reads, tasks, api.default_finish_params(flank), ext_stride=768)
    reads.free()
    common.same_finish_output(got, want)
    assert [("%.1f" % v) for v in want[0]["score"]] == [("%.1f" % v) for v in got[0]["score"]]


def test_scoring_is_per_context(ctx, oracle):
    """Two contexts on one device keep their own scoring_t; setting one does not leak into the other."""
    codes, packed, (blob, off, lens), tasks = common.distance_workload(150, 13, 1000, seed=77, encoding=1, W=177)
    c2 = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    try:
        c2.set_scoring(common.scoring("asym", api))
        _, wd, _ = oracle.align(packed, blob, off, lens, 1, tasks, 0, 1, cigar_stride=512)
        _, wa, _ = oracle.align(packed, blob, off, lens, 1, tasks, 0, 1, scoring=common.scoring("asym"), cigar_stride=512)
        r1 = ctx.upload_reads(blob, off, lens, 1)
        r2 = c2.upload_reads(blob, off, lens, 1)
        g1, _ = ctx.align(r1, tasks, 0, 1, cigar_stride=512)
        g2, _ = c2.align(r2, tasks, 0, 1, cigar_stride=512)
        r1.free()
        r2.free()
        np.testing.assert_array_equal(g1["affine_score"], wd["affine_score"])
        np.testing.assert_array_equal(g2["affine_score"], wa["affine_score"])
    finally:
        c2.destroy()
