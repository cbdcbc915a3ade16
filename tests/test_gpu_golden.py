"""HIP path (through the C-ABI) against the committed fixtures = outputs of the
reference's own compiled objects."""
import numpy as np
import pytest

import golden_util as gu
from whisper_amd import api

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("name", gu.cases("align"))
def test_align(ctx, name):
    z = gu.load(name)
    reads = ctx.upload_reads(z["blob"], z["off"], z["lens"], 1)
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
