"""Non-default scoring_t values (the reference's -match / -mismatch / -gap-* / -clipping flags,
mapper/params.h:111-192): the CPU restatement against the reference's own objects, every routine of the
path that reads a score - getExtCigar + refine_ext_cigar (LevMyers64.cpp:392-520, sam_matching.cpp:858-924),
Match and GetExtCigarNew (indels.cpp:125-461, 680-957: the negative-length insertion quirk moves with
gap_ins_extend), clipBoundaryPositions / clipIllegalPositions (soft_clipping.cpp:592-994), CSoftClipping's
getExtCigar (498-589) and the finish chain with the mapping_desc_t constructor's indel_to_mismatch_score
(common/types.h:342-425).  Scores are compared for equality."""
import numpy as np
import pytest

import common

NAMES = ["asym", "odd"]


@pytest.mark.parametrize("sc_name", NAMES)
@pytest.mark.parametrize("m,k,W,refine", [(150, 13, 177, 1), (150, 26, 660, 1), (100, 10, None, 0), (300, 20, None, 1)])
def test_align(oracle, reference, sc_name, m, k, W, refine):
    sc = common.scoring(sc_name)
    codes, packed, (blob, off, lens), tasks = common.distance_workload(m, k, 2500, seed=3 * m + k, encoding=1, W=W)
    _, a, ca = oracle.align(packed, blob, off, lens, 1, tasks, 0, refine, scoring=sc)
    _, b, cb = reference.align(packed, codes.numel(), blob, off, lens, 1, tasks, 0, refine, scoring=sc)
    _, d, _ = reference.align(packed, codes.numel(), blob, off, lens, 1, tasks, 0, refine)
    assert (b["affine_score"] != d["affine_score"]).sum() > 500      # the scoring does reach the result
    for f in a.dtype.names:
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)
    assert (ca == cb).all()


@pytest.mark.parametrize("sc_name", NAMES)
def test_align_planted_refinement(oracle, reference, sc_name):
    sc = common.scoring(sc_name)
    codes, packed, (blob, off, lens), tasks = common.planted_refine_workload(600, 150, 12, seed=98)
    _, a, ca = oracle.align(packed, blob, off, lens, 1, tasks, 0, 1, scoring=sc)
    _, b, cb = reference.align(packed, codes.numel(), blob, off, lens, 1, tasks, 0, 1, scoring=sc)
    assert b["refined"].sum() > 100
    for f in a.dtype.names:
        np.testing.assert_array_equal(a[f], b[f], err_msg=f)
    assert (ca == cb).all()


@pytest.mark.parametrize("sc_name", NAMES)
@pytest.mark.parametrize("enc,mcf,m,max_indel", [(0, 1.0, 150, 50), (1, 0.6, 150, 50), (0, 0.6, 300, 50), (0, 0.6, 150, 100)])
def test_indel_match_and_ext_cigar(oracle, reference, sc_name, enc, mcf, m, max_indel):
    from oracle import bindings
    sc = common.scoring(sc_name)
    codes, packed, (blob, off, lens), tasks = common.indel_workload(2000, m, seed=700 + m + enc + max_indel, encoding=enc,
                                                                   stage_major=5 if m <= 250 else 8, max_indel=max_indel,
                                                                   max_mm=12 if m <= 250 else 20)
    prm = bindings.default_indel_params(mcf)
    prm.scoring = sc
    _, a = oracle.indel_match(packed, blob, off, lens, enc, tasks, prm)
    _, b = reference.indel_match(packed, blob, off, lens, enc, tasks, prm)
    _, d = reference.indel_match(packed, blob, off, lens, enc, tasks, bindings.default_indel_params(mcf))
    assert b["found"].sum() > 500 and (b != d).sum() > 3            # other costs pick other mappings
    np.testing.assert_array_equal(a, b)
    ct = common.indel_cigar_tasks(tasks, b)
    _, ra, ca = oracle.indel_ext_cigar(packed, blob, off, lens, enc, ct, scoring=sc, cigar_stride=2048)
    _, rb, cb = reference.indel_ext_cigar(packed, blob, off, lens, enc, ct, max_indel=max_indel, scoring=sc,
                                          cigar_stride=2048)
    np.testing.assert_array_equal(ra, rb)
    np.testing.assert_array_equal(ca, cb)
    ins = ct["indel_len1"] < 0
    assert ins.sum() > 50                                             # the insertion-score quirk is exercised


@pytest.mark.parametrize("sc_name", NAMES)
def test_clip_boundary_and_illegal(oracle, reference, sc_name):
    sc = common.scoring(sc_name)
    rec, cig = common.clip_workload(2500, 150, seed=18, scoring=sc)
    _, ra, ca = oracle.clip_boundary(rec, cig, sc)
    _, rb, cb = reference.clip_boundary(rec, cig, sc)
    _, rd, _ = reference.clip_boundary(rec, cig)
    assert (cb != cig).any(axis=1).sum() > 800 and (rb["score"] != rd["score"]).sum() > 500
    for f in ra.dtype.names:
        if f != "no_clippings":
            np.testing.assert_array_equal(ra[f], rb[f], err_msg=f)
    np.testing.assert_array_equal(ca, cb)
    t, rec, cig = common.clip_illegal_workload(oracle, 2500, seed=6, scoring=sc)
    rc1, ra, ca = oracle.clip_illegal(t, rec, cig, sc)
    rc2, rb, cb = reference.clip_illegal(t, rec, cig, sc)
    assert rc1 == 0 and rc2 == 0 and (ra["no_clippings"] > 0).sum() > 1000
    for f in ra.dtype.names:
        if f != "reserved":
            np.testing.assert_array_equal(ra[f], rb[f], err_msg=f)
    for i in range(len(ra)):
        n = ra["cigar_len"][i] + 1
        assert bytes(ca[i, :n]) == bytes(cb[i, :n]), i


@pytest.mark.parametrize("sc_name", NAMES)
def test_softclip_match(oracle, reference, sc_name):
    from oracle import bindings as ob
    sc = common.scoring(sc_name)
    codes, packed, (blob, off, lens), tasks = common.softclip_workload(1500, seed=713, W=700, encoding=1)
    prm = ob.default_softclip_params(fresh_object=0)
    rc1, a, ca = oracle.softclip_match(packed, blob, off, lens, 1, tasks, prm, sc)
    rc2, b, cb = reference.softclip_match(packed, blob, off, lens, 1, tasks, prm, sc)
    assert rc1 == 0 and rc2 == 0 and a["found"].sum() > 1000
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ca, cb)


@pytest.mark.parametrize("sc_name", NAMES)
@pytest.mark.parametrize("seed,m", [(25, 150), (26, 100)])
def test_finish_pairs_chain(oracle, reference, sc_name, seed, m):
    sc = common.scoring(sc_name)
    (codes, packed), (blob, off, lens), desc, ridx, maps, direct, distant, stale = common.finish_workload(
        oracle, 500, seed, m=m, scoring=sc)
    _, wire, _ = oracle.serialize(maps, direct)
    rc, tasks, res, ext, cig, mdz = reference.finish_pairs(packed, codes.numel(), desc, blob, off, lens, ridx, wire,
                                                           distant, stale, scoring=sc, ext_stride=768)
    assert rc == 0 and res["ok"].all()
    mine = common.finish_tasks(desc, ridx, lens, maps, direct, distant, stale, scoring=sc)
    for f in tasks.dtype.names:
        np.testing.assert_array_equal(mine[f], tasks[f], err_msg=f)
    rc, *got = oracle.finish_pairs(packed, blob, off, lens, tasks, scoring=sc, ext_stride=768)
    assert rc == 0
    common.same_finish_output(got, (res, ext, cig, mdz))
    meth = np.bincount(res["method"], minlength=27)
    assert meth[2] + meth[3] > 400 and meth[13:26].sum() > 20
    assert (res["left_clipping"] > 0).sum() > 30 and (res["right_clipping"] > 0).sum() > 30
