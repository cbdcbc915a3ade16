"""The CPU restatement (oracle/wv_oracle.c) against the committed fixtures, which hold the
outputs of the reference's own compiled objects (tests/golden/make_golden.py)."""
import numpy as np
import pytest

import golden_util as gu


def _ob():
    from oracle import bindings
    return bindings


@pytest.mark.parametrize("name", gu.cases("distance"))
def test_distance(oracle, name):
    packed, _ = gu.load_genome()
    z = gu.load(name)
    rc, res = oracle.distance(packed, z["blob"], z["off"], z["lens"], int(z["encoding"]), z["tasks"],
                              int(z["last_symbol"]))
    assert rc == 0
    np.testing.assert_array_equal(res, z["results"])


@pytest.mark.parametrize("name", gu.cases("levdiag"))
def test_levdiag(oracle, name):
    packed, ref_size = gu.load_genome()
    z = gu.load(name)
    rc, res = oracle.levdiag(packed, ref_size, int(z["cur_offset"]), z["blob"], z["off"], z["lens"],
                             z["tasks"])
    assert rc == 0
    np.testing.assert_array_equal(res, z["results"])


@pytest.mark.parametrize("name", gu.cases("align"))
def test_align(oracle, name):
    packed, _ = gu.load_genome()
    z = gu.load(name)
    rc, res, cig = oracle.align(packed, z["blob"], z["off"], z["lens"], 1, z["tasks"],
                                int(z["last_symbol"]), int(z["refine"]),
                                cigar_stride=int(z["cigar_stride"]), scoring=gu.scoring(z, _ob()))
    assert rc == 0
    want = z["results"]
    for f in want.dtype.names:   # affine_score included: same doubles added in the same order
        np.testing.assert_array_equal(res[f], want[f], err_msg=f)
    for i in np.nonzero(want["ok"])[0]:
        assert gu.cigar_bytes(cig, i) == gu.cigar_bytes(z["cigars"], i), i


@pytest.mark.parametrize("name", gu.cases("indel"))
def test_indel_match_cigar_and_records(oracle, name):
    from oracle import bindings
    packed, _ = gu.load_genome()
    z = gu.load(name)
    enc = int(z["encoding"])
    prm = bindings.default_indel_params(float(z["min_clipped_factor"]))
    sc = gu.scoring(z, bindings)
    if sc is not None:
        prm.scoring = sc
    rc, res = oracle.indel_match(packed, z["blob"], z["off"], z["lens"], enc, z["tasks"], prm)
    np.testing.assert_array_equal(res, z["results"])
    rc, cres, ccig = oracle.indel_ext_cigar(packed, z["blob"], z["off"], z["lens"], enc, z["cigar_tasks"],
                                            scoring=sc, cigar_stride=int(z["cigar_stride"]))
    np.testing.assert_array_equal(cres, z["cigar_results"])
    np.testing.assert_array_equal(ccig, z["cigars"])
    rc, wire, sizes = oracle.serialize(z["results"], z["direction"])
    np.testing.assert_array_equal(wire, z["wire"])
    np.testing.assert_array_equal(sizes, z["wire_sizes"])


@pytest.mark.parametrize("name", gu.cases("clip"))
def test_clip_boundary(oracle, name):
    z = gu.load(name)
    rc, rec, cig = oracle.clip_boundary(z["records"], z["cigars"], gu.scoring(z, _ob()))
    want = z["out_records"]
    for f in want.dtype.names:
        if f != "no_clippings":      # not a field of the reference's mapping_desc_t
            np.testing.assert_array_equal(rec[f], want[f], err_msg=f)
    np.testing.assert_array_equal(cig, z["out_cigars"])


def test_ext_cigar_checker_accepts_oracle_output(oracle):
    """The replay checker the large GPU test relies on: consistent with the oracle's own
    alignments, and it notices a corrupted CIGAR."""
    import common
    codes, packed, (blob, off, lens), tasks = common.distance_workload(150, 13, 3000, seed=99, encoding=1, W=177,
                                                                      true_frac=0.9)
    rc, res, cig = oracle.align(packed, blob, off, lens, 1, tasks, last_symbol=0, refine=1, cigar_stride=512)
    assert rc == 0 and res["ok"].sum() > 1500
    fail = oracle.check_ext_cigars(packed, blob, off, lens, tasks, res, cig)
    assert not fail.any()
    i = int(np.nonzero(res["ok"] & (res["cigar_len"] > 20))[0][0])
    cig2 = cig.copy()
    cig2[i, 10] = ord("#") if cig2[i, 10] == ord(".") else ord(".")
    fail = oracle.check_ext_cigars(packed, blob, off, lens, tasks, res, cig2)
    assert fail[i] != 0 and not np.delete(fail, i).any()


@pytest.mark.parametrize("name", gu.cases("cf"))
def test_cf_filter(oracle, name):
    z = gu.load(name)
    packed, ref_size = gu.load_genome()
    rc, res = oracle.cf_filter(packed, ref_size, int(z["cur_offset"]), z["blob"], z["off"], z["lens"], z["tasks"])
    assert rc == 0
    np.testing.assert_array_equal(res, z["results"])
    np.testing.assert_array_equal((res <= int(z["max_diff"])).astype(np.uint8), z["passed"])

def _check_sam_cigar(z, cig, mdz, cbin, lens):
    np.testing.assert_array_equal(lens, z["lens"])
    for i in range(len(lens)):
        a, b, c = int(lens["cigar_len"][i]), int(lens["mdz_len"][i]), int(lens["cigar_bin_len"][i])
        assert bytes(cig[i, :a]) == bytes(z["cigars"][i, :a])
        assert bytes(mdz[i, :b]) == bytes(z["mdz"][i, :b])
        np.testing.assert_array_equal(cbin[i, :c], z["cigar_bin"][i, :c])


@pytest.mark.parametrize("name", gu.cases("samcigar"))
def test_sam_cigar(oracle, name):
    z = gu.load(name)
    rc, cig, mdz, cbin, lens = oracle.sam_cigar(z["ext"])
    assert rc == 0
    _check_sam_cigar(z, cig, mdz, cbin, lens)


def _check_clip_l(z, rec, cig, skip):
    want = z["out_records"]
    for f in want.dtype.names:
        if f not in ("reserved",) + skip:
            np.testing.assert_array_equal(rec[f], want[f], err_msg=f)
    for i in range(len(rec)):
        assert gu.cigar_bytes(cig, i) == gu.cigar_bytes(z["out_cigars"], i), i


@pytest.mark.parametrize("name", gu.cases("clipillegal"))
def test_clip_illegal(oracle, name):
    z = gu.load(name)
    rc, rec, cig = oracle.clip_illegal(z["tasks"], z["records"], z["cigars"], gu.scoring(z, _ob()))
    assert rc == 0
    _check_clip_l(z, rec, cig, ())


@pytest.mark.parametrize("name", gu.cases("clippairs"))
def test_clip_overlapping_pairs(oracle, name):
    z = gu.load(name)
    rc, rec, cig = oracle.clip_overlapping_pairs(z["tasks"], z["records"], z["cigars"])
    assert rc == 0
    _check_clip_l(z, rec, cig, ("no_clippings",))


@pytest.mark.parametrize("name", gu.cases("softclip"))
def test_softclip_match(oracle, name):
    from oracle import bindings as ob
    z = gu.load(name)
    packed, _ = gu.load_genome()
    prm = ob.default_softclip_params(int(z["max_query_len"]), int(z["max_mismatches"]), int(z["fresh_object"]))
    rc, res, cig = oracle.softclip_match(packed, z["blob"], z["off"], z["lens"], int(z["encoding"]), z["tasks"], prm,
                                         gu.scoring(z, ob), cigar_stride=int(z["cigar_stride"]))
    assert rc == 0
    np.testing.assert_array_equal(res, z["results"])
    np.testing.assert_array_equal(cig, z["cigars"])


@pytest.mark.parametrize("name", gu.cases("finish"))
def test_finish_pairs(oracle, name):
    """The chain fixture: descriptors in, SAM fields out, as the reference's own objects produced them."""
    import common
    from oracle import bindings
    packed, _ = gu.load_genome()
    z = gu.load(name)
    prm = bindings.default_finish_params(int(z["flank"]))
    rc, *got = oracle.finish_pairs(packed, z["blob"], z["off"], z["lens"], z["tasks"], prm, gu.scoring(z, bindings),
                                   ext_stride=int(z["ext_stride"]))
    assert rc == 0
    common.same_finish_output(got, (z["results"], z["ext_cigars"], z["cigars"], z["mdz"]))
    # the descriptors themselves follow from the wire records and the chromosome table
    maps_dir = z["wire"][:, 4] >> 7
    assert ((z["tasks"]["direct"] == maps_dir)).all()


@pytest.mark.parametrize("name", gu.cases("groups"))
def test_commit_and_result_groups(oracle, name):
    from oracle import bindings
    z = gu.load(name)
    rc, committed = oracle.commit(z["tasks"], z["indel"])
    assert rc == 0
    np.testing.assert_array_equal(committed, z["committed"])
    prm = bindings.GroupParams(int(z["id_bytes"]), int(z["mapping_counter_size"]), int(z["max_no_mappings"]), 0)
    rc, data = oracle.result_groups(z["tasks"], committed, z["read_first"], z["read_ids"], prm)
    assert rc == 0 and bytes(data) == bytes(z["group_bytes"])


@pytest.mark.parametrize("name", gu.cases("saloop"))
def test_sa_loops(oracle, name):
    """The reference's own suffix-array loops (fixture: what CMappingCore::search_in_unknown_range_dir_and_rc /
    search_in_known_range_dir_and_rc pushed, and the bytes its heap gatherer + collector made of that) against the loop
    body composed from the pinned oracle functions - the composition the device entry points implement."""
    from oracle import bindings as ob
    z = gu.load(name)
    packed, ref_size, k, fl = z["packed"], int(z["ref_size"]), int(z["k"]), int(z["seg"][0])
    blob, off, lens = z["blob"], z["off"], z["lens"]
    rc, val = oracle.cf_filter(packed, ref_size, fl, blob, off, lens, z["cf"])
    passed = (z["cf"]["pattern_len"] == 0) | (val <= 4 * k)
    if k < 7:      # lev_alg_thr, common/defs.h:108
        rc, ed = oracle.levdiag(packed, ref_size, fl, blob, off, lens, z["lev"])
    else:
        rc, res = oracle.distance(packed, blob, off, lens, 0, z["dist"], -1)
        ed = res["edit_distance"]
    commit = z["commit"].copy()
    commit["lev_errors"] = np.where(passed, ed, k + 1)
    im = None
    if int(z["sensitive"]):
        commit["has_indel"] = 1
        rc, im = oracle.indel_match(packed, blob, off, lens, 0, z["indel"], ob.default_indel_params(float(z["min_clipped_factor"])))
    rc, committed = oracle.commit(commit, im)
    kept = committed["found"] == 1
    for f in ("type", "pos", "no_mismatches", "matching_len1", "matching_len2", "indel_len1", "indel_len2", "clipping_len",
              "penalty"):
        np.testing.assert_array_equal(committed[f][kept], z["pushed"][f], err_msg=f)
    np.testing.assert_array_equal(commit["direct"][kept], z["pushed_direct"])
    rc, group = oracle.result_groups(commit, committed, z["read_first"], z["read_ids"], ob.GroupParams(5, 4, 1024, 0))
    assert rc == 0 and bytes(group) == bytes(z["group_bytes"])
