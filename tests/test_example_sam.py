"""BASELINE configs[0]: the reference's own golden output for its example reads (example/out-se.sam, out-pe.sam over
SRR10051130_10k_{1,2}.fastq - what its CI diffs against, .github/workflows/main.yml:29-45) reproduced record by
record: POS, CIGAR, MD:Z, NM:i and the AS:i string of every mapped record whose alignment can be re-run on the
reference spans its own SEQ + CIGAR + MD:Z spell out (tests/example_sam.py, tests/golden/make_example_golden.py).
4 447 records: 2 362 single-end (fill_cigar_with_lev / _indel_matching), 2 085 paired-end (perfect, lev, mate rescue
by LevMyers, mate rescue by hashing, long-indel records; 95 of them soft-clipped, 1 037 pairs through
clipOverlappingPairs).  3 records are excluded: the reference's own objects do not reproduce them on the planted
windows (their alignment used the real flank)."""
import numpy as np
import pytest

import example_sam as ex
import golden_util as gu


def _load():
    z = np.load(gu.GOLDEN + "/example_sam.npz")
    strs = [[bytes(a[i]).split(b"\0")[0] for i in range(len(a))] for a in (z["cigar"], z["md"], z["as_"])]
    return z, z["recs"], strs


def _check(be, z, recs, strs, label):
    be.set_reads(z["blob"], z["off"], z["lens"])
    got = ex.run_chain(be, recs, z["lens"], pairs=True)
    same = ex.compare(recs, strs[0], strs[1], strs[2], got)
    keep = recs["excluded"] == 0
    bad = np.nonzero(~same & keep)[0]
    assert len(bad) == 0, "%s: record %d: CIGAR %r MD:Z %r NM %d AS %s POS %d, the SAM has %r %r %d %r %d" % (
        label, bad[0], got[1][bad[0]], got[2][bad[0]], got[0]["err_edit_distance"][bad[0]],
        ex.double2pchar(float(got[0]["score"][bad[0]])), got[0]["ref_seq_pos"][bad[0]], strs[0][bad[0]], strs[1][bad[0]],
        recs["nm"][bad[0]], strs[2][bad[0]], recs["pos"][bad[0]])
    return got


def test_fixture_covers_the_example_run():
    z, recs, strs = _load()
    src = z["source"]
    assert len(recs) == 4447 and (src == 0).sum() == 2362 and int(recs["excluded"].sum()) == 3
    kinds = np.bincount(recs["kind"][src == 1], minlength=5)
    assert list(kinds) == [1001, 1016, 53, 13, 2]
    clipped = np.array([b"S" in c for c in strs[0]])
    assert (clipped & (recs["excluded"] == 0)).sum() == 92
    assert (recs["mate"] >= 0).sum() == 2 * 1037
    assert all(len(bytes(z["blob"][int(o):int(o) + int(l)])) == 151 for o, l in zip(z["off"][:50], z["lens"][:50]))


def test_as_field_formatting():
    """SDouble2PChar (common/utils.h:276-286)."""
    assert ex.double2pchar(145.0) == "145.0" and ex.double2pchar(136.6) == "136.6" and ex.double2pchar(109.8) == "109.8"
    assert ex.double2pchar(12.25) == "12.3" and ex.double2pchar(0.0) == "0.0"


def test_oracle_reproduces_the_reference_sam(oracle):
    z, recs, strs = _load()
    _check(ex.OracleBackend(oracle, z["packed"], int(z["ref_size"]), False), z, recs, strs, "oracle")


def test_reference_objects_reproduce_their_own_sam(reference):
    z, recs, strs = _load()
    _check(ex.OracleBackend(reference, z["packed"], int(z["ref_size"]), True), z, recs, strs, "reference")


@pytest.mark.gpu
def test_gpu_reproduces_the_reference_sam(oracle):
    """Every stage through the C-ABI: wamd_align / wamd_softclip_match / wamd_indel_ext_cigar, wamd_clip_illegal,
    wamd_clip_boundary, wamd_clip_overlapping_pairs, wamd_sam_cigar."""
    from whisper_amd import api
    z, recs, strs = _load()
    ctx = api.Context(z["packed"], ref_size=int(z["ref_size"]), cur_offset=0)
    be = ex.GpuBackend(ctx)
    try:
        got = _check(be, z, recs, strs, "gpu")
        # and on the excluded records the HIP path still equals the CPU restatement
        orc = ex.OracleBackend(oracle, z["packed"], int(z["ref_size"]), False)
        orc.set_reads(z["blob"], z["off"], z["lens"])
        want = ex.run_chain(orc, recs, z["lens"], pairs=True)
        for f in ("score", "ref_seq_pos", "ref_length", "err_edit_distance", "left_clipping", "right_clipping"):
            np.testing.assert_array_equal(got[0][f], want[0][f], err_msg=f)
        assert got[1] == want[1] and got[2] == want[2]
    finally:
        be.close()
        ctx.destroy()


@pytest.mark.gpu
def test_gpu_finish_pairs_reproduces_the_paired_end_sam():
    """The paired-end records the reference finished at sam.cpp:1871-1900 in one go - both mates perfect or lev,
    clipBoundaryPositions applied once - through wamd_finish_pairs: mapping record in, SAM columns out."""
    from whisper_amd import api
    z, recs, strs = _load()
    lens = z["lens"]
    first = np.nonzero((recs["mate"] >= 0) & (recs["first"] == 1) & (recs["excluded"] == 0))[0]
    sel = []
    for i in first:
        j = recs["mate"][i]
        if all(recs["kind"][x] in (ex.KIND_PERFECT, ex.KIND_LEV) and recs["n_boundary"][x] == 1 and recs["excluded"][x] == 0
               for x in (i, j)):
            sel += [i, j]
    sel = np.array(sel)
    assert len(sel) > 1500
    r = recs[sel]
    m = lens[r["read_idx"]].astype(np.int64)
    t = np.zeros(len(sel), api.FINISH_TASK)
    t["read_idx"] = r["read_idx"]
    lev = r["kind"] == ex.KIND_LEV
    begin = np.where(lev, r["ref_pos"].astype(np.int64) + ex.FLANK, r["ref_pos"].astype(np.int64))
    t["pos"] = np.where(r["direct"] == 1, begin, begin + m - 1)
    t["type"] = 0          # WAMD_MAP_LEV
    t["err_edit_distance"] = np.where(lev, r["k"], 0)
    t["ref_seq_pos"] = np.where(lev, r["coord0"] + ex.FLANK, r["coord0"])
    t["ref_size"] = ex.REF_SIZE
    t["direct"] = r["direct"]
    ctx = api.Context(z["packed"], ref_size=int(z["ref_size"]), cur_offset=0)
    try:
        reads = ctx.upload_reads(z["blob"], z["off"], lens, api.READS_ASCII, orientations=3)
        res, ext, cig, mdz = ctx.finish_pairs(reads, t, api.default_finish_params(ex.FLANK), ext_stride=512)
        reads.free()
    finally:
        ctx.destroy()
    assert res["ok"].all()
    for q, i in enumerate(sel):
        got = (bytes(cig[q]).split(b"\0")[0], bytes(mdz[q]).split(b"\0")[0], int(res["err_edit_distance"][q]),
               ex.double2pchar(float(res["score"][q])).encode(), int(res["ref_seq_pos"][q]))
        want = (strs[0][i], strs[1][i], int(recs["nm"][i]), strs[2][i], int(recs["pos"][i]))
        assert got == want, (i, got, want)
