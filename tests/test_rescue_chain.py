"""find_pairs<LevMyers>'s chain per hit (sam_matching.cpp:379-457) as one call, wamd_rescue_mates: the oracle against the
reference's own statements on its own objects (oracle/ref_shim.cpp: ref_rescue_mates_batch), the fixtures made from
those, and the HIP path against both."""
import numpy as np
import pytest

import common
import golden_util as gu


def _same(got, want, gext, wext, label=""):
    for f in want.dtype.names:
        np.testing.assert_array_equal(got[f], want[f], err_msg=label + f)
    for i in np.nonzero(want["ok"])[0]:
        e = want["ext_cigar_len"][i] + 1
        assert bytes(gext[i, :e]) == bytes(wext[i, :e]), i
    assert not gext[want["ok"] == 0, 0].any()


def _stats(res):
    ok = res["ok"] == 1
    assert ok.sum() > len(res) // 3 and (res["aligned"] == 1).sum() > ok.sum() + len(res) // 10      # tests that reject
    assert ((res["left_clipping"] > 0) | (res["right_clipping"] > 0))[ok].sum() > len(res) // 30    # clipped mates


@pytest.mark.parametrize("seed,m,name", [(3, 150, None), (4, 100, "asym"), (5, 250, "odd")])
def test_oracle_equals_the_reference_chain(oracle, reference, seed, m, name):
    from oracle import bindings as ob
    sc = common.scoring(name) if name else ob.default_scoring()
    (codes, packed), (blob, off, lens), desc, mate, hit_pos, seq_id, events, direct, k = common.rescue_workload(1200, seed, m)
    rc, tasks, want, wext = reference.rescue_mates(packed, codes.numel(), desc, blob, off, lens, mate, hit_pos, seq_id, events,
                                                   direct, k, scoring=sc, model=common.RESCUE_MODEL, ext_stride=768)
    assert rc == 0
    _stats(want)
    rc, got, gext = oracle.rescue_mates(packed, blob, off, lens, tasks, scoring=sc, ext_stride=768)
    assert rc == 0
    _same(got, want, gext, wext)
    # the switches of CParams
    for sw in ((0, 1), (1, 0)):
        prm = ob.RescueParams(*sw)
        rc, tasks, want, wext = reference.rescue_mates(packed, codes.numel(), desc, blob, off, lens, mate[:400], hit_pos[:400],
                                                       seq_id[:400], events[:400], direct[:400], k[:400], params=prm,
                                                       scoring=sc, model=common.RESCUE_MODEL, ext_stride=768)
        rc, got, gext = oracle.rescue_mates(packed, blob, off, lens, tasks, params=prm, scoring=sc, ext_stride=768)
        _same(got, want, gext, wext, "switch %s " % (sw,))


@pytest.mark.parametrize("name", gu.cases("rescue"))
def test_oracle_reproduces_the_fixtures(oracle, name):
    z = gu.load(name)
    from oracle import bindings as ob
    rc, got, gext = oracle.rescue_mates(z["packed"], z["blob"], z["off"], z["lens"], z["tasks"], scoring=gu.scoring(z, ob),
                                        ext_stride=512)
    assert rc == 0
    _same(got, z["results"], gext, z["ext"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", gu.cases("rescue"))
def test_gpu_reproduces_the_fixtures(name):
    from whisper_amd import api
    z = gu.load(name)
    ctx = api.Context(z["packed"], ref_size=int(z["ref_size"]), cur_offset=0)
    try:
        ctx.set_scoring(gu.scoring(z, api) or api.default_scoring())
        reads = ctx.upload_reads(z["blob"], z["off"], z["lens"], api.READS_ASCII, orientations=3)
        got, gext = ctx.rescue_mates(reads, z["tasks"], ext_stride=512)
        reads.free()
    finally:
        ctx.destroy()
    _same(got, z["results"], gext, z["ext"])


@pytest.mark.gpu
@pytest.mark.parametrize("seed,m,name", [(13, 150, None), (14, 250, "asym"), (15, 100, None)])
def test_gpu_equals_the_oracle(oracle, seed, m, name):
    from oracle import bindings as ob
    from whisper_amd import api
    sc = common.scoring(name) if name else ob.default_scoring()
    (codes, packed), (blob, off, lens), desc, mate, hit_pos, seq_id, events, direct, k = common.rescue_workload(3000, seed, m)
    tasks = common.rescue_tasks(desc, mate, hit_pos, seq_id, events, direct, k)
    ctx = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    try:
        ctx.set_scoring(common.scoring(name, api) if name else api.default_scoring())
        reads = ctx.upload_reads(blob, off, lens, api.READS_ASCII, orientations=3)
        for sw in ((1, 1), (0, 1), (1, 0)):
            rc, want, wext = oracle.rescue_mates(packed, blob, off, lens, tasks, params=ob.RescueParams(*sw), scoring=sc,
                                                 ext_stride=768)
            assert rc == 0
            got, gext = ctx.rescue_mates(reads, tasks, params=api.RescueParams(*sw), ext_stride=768)
            _same(got, want, gext, wext, "switch %s " % (sw,))
        _stats(want)
        reads.free()
    finally:
        ctx.destroy()
