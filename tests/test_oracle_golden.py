"""The CPU restatement (oracle/wv_oracle.c) against the committed fixtures, which hold the
outputs of the reference's own compiled objects (tests/golden/make_golden.py)."""
import numpy as np
import pytest

import golden_util as gu


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
                                cigar_stride=int(z["cigar_stride"]))
    assert rc == 0
    want = z["results"]
    for f in want.dtype.names:   # affine_score included: same doubles added in the same order
        np.testing.assert_array_equal(res[f], want[f], err_msg=f)
    for i in np.nonzero(want["ok"])[0]:
        assert gu.cigar_bytes(cig, i) == gu.cigar_bytes(z["cigars"], i), i
