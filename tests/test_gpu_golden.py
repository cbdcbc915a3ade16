"""HIP path (through the C-ABI) against the committed fixtures = outputs of the
reference's own compiled objects."""
import contextlib

import numpy as np
import pytest

import golden_util as gu
from whisper_amd import api

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def scored(ctx, z):
    """Fixtures generated under a non-default scoring_t carry it: wamd_ctx_set_scoring for the test's calls."""
    sc = gu.scoring(z, api)
    if sc is not None:
        ctx.set_scoring(sc)
    try:
        yield sc
    finally:
        if sc is not None:
            ctx.set_scoring(api.default_scoring())


@pytest.fixture(scope="module")
def ctx():
    packed, ref_size = gu.load_genome()
    c = api.Context(packed, ref_size=ref_size, cur_offset=0)
    yield c
    c.destroy()


@pytest.mark.parametrize("name", gu.cases("distance"))
def test_distance(ctx, name):
    z = gu.load(name)
    reads = ctx.upload_reads(z["blob"], z["off"], z["lens"], int(z["encoding"]))
    got = ctx.dynamic_programming(reads, z["tasks"], int(z["last_symbol"]))
    reads.free()
    np.testing.assert_array_equal(got, z["results"])


@pytest.mark.parametrize("name", gu.cases("levdiag"))
def test_levdiag(ctx, name):
    z = gu.load(name)
    reads = ctx.upload_reads(z["blob"], z["off"], z["lens"], 0, orientations=1)
    got = ctx.lev_diag(reads, z["tasks"])
    reads.free()
    np.testing.assert_array_equal(got, z["results"])


@pytest.mark.parametrize("name", gu.cases("cf"))
def test_cf_filter(ctx, name):
    z = gu.load(name)
    _, ref_size = gu.load_genome()
    ctx.set_reference_params(ref_size, int(z["cur_offset"]))
    try:
        reads = ctx.upload_reads(z["blob"], z["off"], z["lens"], 0, orientations=1)
        got = ctx.cf_filter(reads, z["tasks"])
        reads.free()
    finally:
        ctx.set_reference_params(ref_size, 0)
    np.testing.assert_array_equal(got, z["results"])
    np.testing.assert_array_equal((got <= int(z["max_diff"])).astype(np.uint8), z["passed"])


@pytest.mark.parametrize("name", gu.cases("align"))
def test_align(ctx, name):
    z = gu.load(name)
    reads = ctx.upload_reads(z["blob"], z["off"], z["lens"], 1)
    with scored(ctx, z):
        got, cig = ctx.align(reads, z["tasks"], int(z["last_symbol"]), int(z["refine"]),
                             cigar_stride=int(z["cigar_stride"]))
    reads.free()
    want = z["results"]
    for f in want.dtype.names:
        if f == "affine_score":   # north_star: within 1e-5; in fact identical
            np.testing.assert_allclose(got[f], want[f], rtol=0, atol=1e-5)
        np.testing.assert_array_equal(got[f], want[f], err_msg=f)
    for i in np.nonzero(want["ok"])[0]:
        assert gu.cigar_bytes(cig, i) == gu.cigar_bytes(z["cigars"], i), i


@pytest.mark.parametrize("name", gu.cases("indel"))
def test_indel_match_cigar_and_records(ctx, name):
    z = gu.load(name)
    enc = int(z["encoding"])
    reads = ctx.upload_reads(z["blob"], z["off"], z["lens"], enc, orientations=1)
    got = ctx.indel_match(reads, z["tasks"], api.default_indel_params(float(z["min_clipped_factor"]), gu.scoring(z, api)))
    np.testing.assert_array_equal(got, z["results"])
    with scored(ctx, z):
        cres, ccig = ctx.indel_ext_cigar(reads, z["cigar_tasks"], cigar_stride=int(z["cigar_stride"]))
    reads.free()
    np.testing.assert_array_equal(cres, z["cigar_results"])
    for i in range(len(cres)):
        assert gu.cigar_bytes(ccig, i) == gu.cigar_bytes(z["cigars"], i), i
    wire, sizes = ctx.serialize_mappings(z["results"], z["direction"])
    np.testing.assert_array_equal(sizes, z["wire_sizes"])
    for i in range(len(sizes)):
        assert bytes(wire[i, :sizes[i]]) == bytes(z["wire"][i, :sizes[i]]), i


@pytest.mark.parametrize("name", gu.cases("clip"))
def test_clip_boundary(ctx, name):
    z = gu.load(name)
    with scored(ctx, z):
        rec, cig = ctx.clip_boundary(z["records"], z["cigars"])
    want = z["out_records"]
    for f in want.dtype.names:
        if f not in ("no_clippings", "reserved"):
            np.testing.assert_array_equal(rec[f], want[f], err_msg=f)
    for i in range(len(rec)):
        assert gu.cigar_bytes(cig, i) == gu.cigar_bytes(z["out_cigars"], i), i

def _check_sam_cigar(z, cig, mdz, cbin, lens):
    np.testing.assert_array_equal(lens, z["lens"])
    for i in range(len(lens)):
        a, b, c = int(lens["cigar_len"][i]), int(lens["mdz_len"][i]), int(lens["cigar_bin_len"][i])
        assert bytes(cig[i, :a]) == bytes(z["cigars"][i, :a])
        assert bytes(mdz[i, :b]) == bytes(z["mdz"][i, :b])
        np.testing.assert_array_equal(cbin[i, :c], z["cigar_bin"][i, :c])


@pytest.mark.parametrize("name", gu.cases("samcigar"))
def test_sam_cigar(ctx, name):
    z = gu.load(name)
    cig, mdz, cbin, lens = ctx.sam_cigar(z["ext"])
    _check_sam_cigar(z, cig, mdz, cbin, lens)


def _check_clip_l(z, rec, cig, skip):
    want = z["out_records"]
    for f in want.dtype.names:
        if f not in ("reserved",) + skip:
            np.testing.assert_array_equal(rec[f], want[f], err_msg=f)
    for i in range(len(rec)):
        assert gu.cigar_bytes(cig, i) == gu.cigar_bytes(z["out_cigars"], i), i


@pytest.mark.parametrize("name", gu.cases("clipillegal"))
def test_clip_illegal(ctx, name):
    z = gu.load(name)
    with scored(ctx, z):
        rec, cig = ctx.clip_illegal(z["tasks"], z["records"], z["cigars"])
    _check_clip_l(z, rec, cig, ())


@pytest.mark.parametrize("name", gu.cases("clippairs"))
def test_clip_overlapping_pairs(ctx, name):
    z = gu.load(name)
    rec, cig = ctx.clip_overlapping_pairs(z["tasks"], z["records"], z["cigars"])
    _check_clip_l(z, rec, cig, ("no_clippings",))


@pytest.mark.parametrize("name", gu.cases("softclip"))
def test_softclip_match(ctx, name):
    z = gu.load(name)
    prm = api.default_softclip_params(int(z["max_query_len"]), int(z["max_mismatches"]), int(z["fresh_object"]))
    reads = ctx.upload_reads(z["blob"], z["off"], z["lens"], int(z["encoding"]), orientations=3)
    with scored(ctx, z):
        res, cig = ctx.softclip_match(reads, z["tasks"], prm, cigar_stride=int(z["cigar_stride"]))
    reads.free()
    np.testing.assert_array_equal(res, z["results"])
    for i in range(len(res)):
        assert gu.cigar_bytes(cig, i) == gu.cigar_bytes(z["cigars"], i), i


@pytest.mark.parametrize("name", gu.cases("finish"))
def test_finish_pairs(ctx, name):
    """mapping records -> ext-CIGAR -> three clipping passes -> CIGAR / MD:Z / NM / AS without leaving the device,
    against the reference's own chain (oracle/ref_shim.cpp: ref_finish_pairs_batch)."""
    import common
    z = gu.load(name)
    reads = ctx.upload_reads(z["blob"], z["off"], z["lens"], 1, orientations=3)
    with scored(ctx, z):
        got = ctx.finish_pairs(reads, z["tasks"], api.default_finish_params(int(z["flank"])), ext_stride=int(z["ext_stride"]))
    reads.free()
    common.same_finish_output(got, (z["results"], z["ext_cigars"], z["cigars"], z["mdz"]))
    # AS:i: is Double2PChar of the score (utils.h:276-286): one decimal, identical because the doubles are
    want, have = z["results"]["score"], got[0]["score"]
    assert [("%.1f" % v) for v in want] == [("%.1f" % v) for v in have]


@pytest.mark.parametrize("name", gu.cases("groups"))
def test_commit_and_result_groups(ctx, name):
    """Verified candidates -> the bytes of a result group, against the reference's construct<lev> / better /
    CMappingsHeapGatherer / CMappingResultsCollector::Push."""
    import common
    z = gu.load(name)
    committed, data, offs, flags, _ = common.gpu_result_groups(ctx, z["tasks"], z["indel"], z["read_first"], z["read_ids"],
                                                               int(z["id_bytes"]), int(z["max_no_mappings"]))
    assert flags == 0
    np.testing.assert_array_equal(committed, z["committed"])
    assert bytes(data) == bytes(z["group_bytes"])
