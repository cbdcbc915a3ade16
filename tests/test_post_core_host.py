"""whisper_amd/csrc/post_core.h - the event-driven per-string routines of rows B / L / X that the HIP kernel runs one
lane each - compiled for the host (tests/native/post_core_host.cpp) and held against the oracle record by record: the
same header and arithmetic as on the device, so what fails here would fail there.  The harness also checks, after every
pass, that the event bitmap a routine maintained by splicing equals a fresh digest of the string it left."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import common

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "native", "libpost_core_host.so")
SRC = os.path.join(HERE, "native", "post_core_host.cpp")
HDR = os.path.join(HERE, "..", "whisper_amd", "csrc", "post_core.h")


@pytest.fixture(scope="module")
def host():
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", SO, SRC])
    return C.CDLL(SO)


def _same(grec, gcig, wrec, wcig):
    for f in wrec.dtype.names:
        if f != "reserved":
            np.testing.assert_array_equal(grec[f], wrec[f], err_msg=f)
    for i in range(len(wrec)):
        n = wrec["cigar_len"][i] + 1
        assert bytes(gcig[i, :n]) == bytes(wcig[i, :n]), (i, bytes(gcig[i, :n]), bytes(wcig[i, :n]))


SCORINGS = [None, "asym", "odd"]


@pytest.mark.parametrize("name", SCORINGS)
@pytest.mark.parametrize("m", [150, 250])
def test_clip_boundary(host, oracle, name, m):
    from oracle import bindings as ob
    sc = common.scoring(name) if name else ob.default_scoring()
    rec, cig = common.clip_workload(2500, m, seed=22 + m, scoring=sc)
    _, wrec, wcig = oracle.clip_boundary(rec, cig, sc)
    grec = np.array(rec, ob.CLIP_RECORD, copy=True)
    gcig = np.array(cig, np.uint8, copy=True)
    rc = host.pc_clip_boundary_batch(C.c_uint32(len(grec)), ob._p(grec), ob._p(gcig), C.c_uint32(gcig.shape[1]), C.byref(sc))
    assert rc == 0 and (wrec["no_clippings"] > 0).sum() > 600
    _same(grec, gcig, wrec, wcig)
    # a second application (the paired-end path applies it once per rescue attempt and again at sam.cpp:1890)
    _, wrec2, wcig2 = oracle.clip_boundary(wrec, wcig, sc)
    rc = host.pc_clip_boundary_batch(C.c_uint32(len(grec)), ob._p(grec), ob._p(gcig), C.c_uint32(gcig.shape[1]), C.byref(sc))
    assert rc == 0
    _same(grec, gcig, wrec2, wcig2)


@pytest.mark.parametrize("name", SCORINGS)
@pytest.mark.parametrize("m", [100, 150])
def test_clip_illegal(host, oracle, name, m):
    from oracle import bindings as ob
    sc = common.scoring(name) if name else ob.default_scoring()
    t, rec, cig = common.clip_illegal_workload(oracle, 2500, seed=35 + m, m=m, scoring=sc)
    rc, wrec, wcig = oracle.clip_illegal(t, rec, cig, sc)
    assert rc == 0 and (wrec["no_clippings"] > 0).sum() > 1200
    got = ob._clip_illegal(host.pc_clip_illegal_batch, t, rec, cig, sc)
    assert got[0] == 0
    _same(got[1], got[2], wrec, wcig)


@pytest.mark.parametrize("m", [100, 150, 250])
def test_clip_pairs(host, oracle, m):
    from oracle import bindings as ob
    t, rec, cig = common.clip_pairs_workload(oracle, 2000, seed=36 + m, m=m)
    rc, wrec, wcig = oracle.clip_overlapping_pairs(t, rec, cig)
    assert rc == 0 and (wrec["no_clippings"] > 0).sum() > 1000
    got = ob._clip_pairs(host.pc_clip_pairs_batch, t, rec, cig)
    assert got[0] == 0
    _same(got[1], got[2], wrec, wcig)


def test_sam_cigar(host, oracle):
    from oracle import bindings as ob
    ext = common.ext_cigar_pool(oracle, n=4000, seed=21)
    rc, wc, wm, wb, wl = oracle.sam_cigar(ext)
    rc, gc, gm, gb, gl = ob._sam_cigar(host.pc_sam_cigar_batch, ext)
    assert rc == 0
    np.testing.assert_array_equal(gl, wl)
    for i in range(len(ext)):
        assert bytes(gc[i, :wl["cigar_len"][i] + 1]) == bytes(wc[i, :wl["cigar_len"][i] + 1]), i
        assert bytes(gm[i, :wl["mdz_len"][i] + 1]) == bytes(wm[i, :wl["mdz_len"][i] + 1]), i
        nb = wl["cigar_bin_len"][i]
        np.testing.assert_array_equal(gb[i, :nb], wb[i, :nb])
    # outputs that cannot fit are refused, never truncated (the oracle, like the reference, has no such bound)
    for stride in (16, 24, 40):
        rc_g, gc, gm, gb, gl = ob._sam_cigar(host.pc_sam_cigar_batch, ext[:600], out_stride=stride)
        over = gl["reserved"] != 0
        assert rc_g == 1 << 6 and over.any()
        assert not gl["cigar_len"][over].any() and not gl["mdz_len"][over].any() and not gc[over, 0].any()
        fits = (wl["cigar_len"][:600] + 12 <= stride) & (wl["mdz_len"][:600] + 14 <= stride)
        assert not over[fits].any()


def test_malformed_strings_are_refused(host):
    from oracle import bindings as ob
    bad = [b"...#", b"..#^A..", b"^", b"....^#C"]
    cig = np.zeros((len(bad), 16), np.uint8)
    for i, b in enumerate(bad):
        cig[i, :len(b)] = np.frombuffer(b, np.uint8)
    full = np.full((1, 16), ord("."), np.uint8)      # no NUL inside the slot
    sc = ob.default_scoring()
    for c in (cig, full):
        for i in range(len(c)):
            rec = np.zeros(1, ob.CLIP_RECORD)
            rec["err_edit_distance"] = 1
            one = c[i:i + 1].copy()
            rc = host.pc_clip_boundary_batch(C.c_uint32(1), ob._p(rec), ob._p(one), C.c_uint32(16), C.byref(sc))
            assert rc == 1 << 5 and bytes(one[0]) == bytes(c[i])


@pytest.mark.parametrize("name,m", [(None, 150), ("odd", 100), ("asym", 250)])
def test_passes_behind_one_another_on_one_digest(host, oracle, name, m):
    """clipIllegalPositions, clipBoundaryPositions, clipOverlappingPairs, extract_cigar_and_mdz in the order of
    sam.cpp:1889-1900 / 2002-2007 on ONE digest per string - the events every pass leaves serve the next."""
    from oracle import bindings as ob
    sc = common.scoring(name) if name else ob.default_scoring()
    pt, rec, cig = common.clip_pairs_workload(oracle, 1500, seed=77 + m, m=m, scoring=sc)
    n = len(rec)
    rec["left_clipping"] = rec["right_clipping"] = 0     # (the workload plants counts without '$' under them)
    rng = np.random.default_rng(5)
    it = np.zeros(n, ob.CLIP_ILLEGAL_TASK)
    it["read_length"] = m
    it["perfect"] = pt["perfect"].reshape(-1)
    last = rec["ref_seq_pos"].astype(np.int64) + rec["ref_length"] - 1
    it["ref_size"] = np.where(rng.random(n) < 0.2, last - rng.integers(1, 30, n), last + 1000)
    early = rng.random(n) < 0.1
    rec["ref_seq_pos"][early] = -rng.integers(0, 30, int(early.sum()))
    rc, r1, c1 = oracle.clip_illegal(it, rec, cig, sc)
    assert rc == 0
    _, r2, c2 = oracle.clip_boundary(r1, c1, sc)
    rc, r3, c3 = oracle.clip_overlapping_pairs(pt, r2, c2)
    assert rc == 0 and (r3["no_clippings"] > 0).sum() > 20 and (r1["no_clippings"] > 0).sum() > 300
    _, wc, wm, _, wl = oracle.sam_cigar(c3)
    grec = np.array(rec, ob.CLIP_RECORD, copy=True)
    gext = np.array(cig, np.uint8, copy=True)
    stride = 2 * gext.shape[1] + 16
    gc, gm = np.zeros((n, stride), np.uint8), np.zeros((n, stride), np.uint8)
    gl = np.zeros(n, ob.SAM_CIGAR_LEN)
    pt_c = np.ascontiguousarray(pt, ob.CLIP_PAIR_TASK)
    rc = host.pc_post_passes(C.c_uint32(n), ob._p(it), ob._p(pt_c), ob._p(grec), ob._p(gext), C.c_uint32(gext.shape[1]),
                             C.byref(sc), 1, 1, ob._p(gc), C.c_uint32(stride), ob._p(gm), C.c_uint32(stride), ob._p(gl))
    assert rc == 0
    _same(grec, gext, r3, c3)
    for f in ("cigar_len", "mdz_len"):
        np.testing.assert_array_equal(gl[f], wl[f])
    for i in range(n):
        assert bytes(gc[i, :wl["cigar_len"][i] + 1]) == bytes(wc[i, :wl["cigar_len"][i] + 1]), i
        assert bytes(gm[i, :wl["mdz_len"][i] + 1]) == bytes(wm[i, :wl["mdz_len"][i] + 1]), i
