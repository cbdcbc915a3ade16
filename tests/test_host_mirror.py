"""The C++ mirror of the reference's verifier objects (whisper_amd/host/gpu_verifier.hpp):
per-read preprocess(), per-candidate dynamicProgramming()/LevDiag() enqueue, one flush.
The replay program is built by `make -C whisper_amd/csrc` (plain g++ over the C-ABI)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "whisper_amd", "host", "replay_batch")


def test_replay_program_is_built():
    assert os.path.exists(BIN), "run `make -C whisper_amd/csrc` (or __graft_entry__.build())"


def _vec(a):
    a = np.ascontiguousarray(a)
    return struct.pack("<Q", a.size) + a.tobytes()


@pytest.mark.gpu
def test_replay_matches_oracle(tmp_path, oracle):
    m, k = 150, 12
    codes, packed, (blob, off, lens), dtasks = common.distance_workload(m, k, 3000, seed=61, rc_frac=0.0)
    fl, fr = common.stage_segments(m, 5)[3]
    # LevDiag tasks on the same reads: same window start p - 5
    ltasks = np.zeros(len(dtasks), common.synth.LEVDIAG_TASK)
    ltasks["read_idx"] = dtasks["read_idx"]
    ltasks["position_in_text"] = dtasks["ref_pos"] + k
    ltasks["fixed_left"], ltasks["fixed_right"] = fl, fr
    ltasks["max_mismatches"] = 5
    ltasks["dir_gen_flag"] = 1
    # The reference's rc branch (mapping_core_sa.cpp:289-291): the pattern is rc_data, handed to
    # preprocess() as genome_t::direct like on the other strand; the window comes from a position in
    # the other strand's suffix array, g0 = ref_size - position - m + cur_offset - k.  Every other
    # task is such a call: same stored symbols, dir_gen_flag = 0, and the distance task's ref_pos
    # written the way :289 computes it.  cur_offset = the anchor's start (mapping_core.cpp:156).
    ref_size, cur_offset = codes.numel(), fl
    p = dtasks["ref_pos"].astype(np.int64) + k
    ltasks["position_in_text"] = p + cur_offset
    rcb = np.arange(len(dtasks)) % 2 == 1
    sa_rc = ref_size - p - m + cur_offset
    ltasks["position_in_text"][rcb] = sa_rc[rcb]
    ltasks["dir_gen_flag"][rcb] = 0
    dtasks["ref_pos"][rcb] = (ref_size - sa_rc - m + cur_offset - k)[rcb]
    _, want_d = oracle.distance(packed, blob, off, lens, 0, dtasks, -1)
    _, want_l = oracle.levdiag(packed, ref_size, cur_offset, blob, off, lens, ltasks)
    assert (want_l[rcb] <= 5).sum() > 100 and (want_l[~rcb] <= 5).sum() > 100
    from oracle import bindings
    if bindings.have_reference():      # the reference's own objects agree on both strands
        ref = bindings.Reference()
        _, ref_d = ref.distance(packed, ref_size, blob, off, lens, 0, dtasks, -1)
        _, ref_l = ref.levdiag(packed, ref_size, cur_offset, blob, off, lens, ltasks)
        np.testing.assert_array_equal(ref_d, want_d)
        np.testing.assert_array_equal(ref_l, want_l)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<4I", ref_size, cur_offset, m, len(lens)))
        f.write(_vec(packed))
        f.write(_vec(blob))
        f.write(_vec(dtasks))
        f.write(_vec(ltasks))
    subprocess.run([BIN, str(fin), str(fout)], check=True, timeout=120)
    raw = np.fromfile(fout, np.uint32)
    got_d = raw[: 2 * len(dtasks)].reshape(-1, 2)
    got_l = raw[2 * len(dtasks):]
    np.testing.assert_array_equal(got_d[:, 0], want_d["edit_distance"])
    np.testing.assert_array_equal(got_d[:, 1], want_d["pos"])
    np.testing.assert_array_equal(got_l, want_l)
    # the switch a mapping stage flips: accepted candidates unchanged, the others {k + 1, 0}
    subprocess.run([BIN, str(fin), str(fout), "--accepted-only"], check=True, timeout=120)
    raw = np.fromfile(fout, np.uint32)
    got_a = raw[: 2 * len(dtasks)].reshape(-1, 2)
    rej = want_d["edit_distance"] > k
    assert rej.any() and (~rej).any()
    np.testing.assert_array_equal(got_a[~rej, 0], want_d["edit_distance"][~rej])
    np.testing.assert_array_equal(got_a[~rej, 1], want_d["pos"][~rej])
    assert (got_a[rej, 0] == k + 1).all() and (got_a[rej, 1] == 0).all()
    np.testing.assert_array_equal(raw[2 * len(dtasks):], want_l)


@pytest.mark.gpu
def test_two_threads_flush_on_one_context(oracle):
    """INTEGRATION section 3 shares one context between the worker threads, each flushing its own
    batches through the host-pointer entry points: the context's lock must keep their staging
    apart (ctypes releases the GIL during a call, so the two threads really overlap)."""
    import threading
    from whisper_amd import api
    codes, packed = common.genome()
    ctx = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    errors = []

    def worker(seed0, sizes):
        try:
            for j, n in enumerate(sizes):
                _, _, (blob, off, lens), tasks = common.distance_workload(150, 12, n, seed=seed0 + j)
                _, want = oracle.distance(packed, blob, off, lens, 0, tasks, -1)
                reads = ctx.upload_reads(blob, off, lens, 0, orientations=3)
                for _ in range(3):
                    got = ctx.dynamic_programming(reads, tasks)
                    if not (got == want).all():
                        errors.append("thread %d batch %d differs" % (seed0, j))
                reads.free()
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    try:
        ts = [threading.Thread(target=worker, args=(500, [3000, 700, 9000, 50, 4000])),
              threading.Thread(target=worker, args=(600, [900, 8000, 60, 5000, 2500]))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    finally:
        ctx.destroy()
    assert not errors, errors


@pytest.mark.gpu
def test_rescue_mirrors_match_oracle(tmp_path, oracle):
    """GpuIndelMatching (CIndelMatching::Preprocess + Match), GpuIndelExtCigar (PreprocessRawSeq + GetExtCigarNew of what
    Match found) and GpuSoftClipping (CSoftClipping::preprocessRawSeq + match + getExtCigar) driven per read / per
    candidate by the replay program."""
    from oracle import bindings as ob
    from whisper_amd import api
    codes, packed, (iblob, ioff, ilen), itasks = common.indel_workload(1500, 150, seed=77)
    order = np.argsort(itasks["read_idx"], kind="stable")
    itasks = itasks[order]
    _, want_i = oracle.indel_match(packed, iblob, ioff, ilen, 0, itasks, ob.default_indel_params(1.0))
    _, _, (sblob, soff, slen), stasks = common.softclip_workload(1200, seed=91, W=700, encoding=1)
    order = np.argsort(stasks["read_idx"], kind="stable")
    stasks = stasks[order]
    sp = ob.default_softclip_params(fresh_object=0)
    _, want_s, want_c = oracle.softclip_match(packed, sblob, soff, slen, 1, stasks, sp)
    stride = want_c.shape[1]
    assert stride % 4 == 0
    assert want_i["found"].sum() > 300 and want_s["found"].sum() > 600
    ct = common.indel_cigar_tasks(itasks, want_i)
    istride = 4 * ((2 * (150 + 2 * 50 + 20) + 8 + 3) // 4)
    _, want_g, want_gc = oracle.indel_ext_cigar(packed, iblob, ioff, ilen, 0, ct, cigar_stride=istride)
    assert len(ct) == want_i["found"].astype(bool).sum() and want_g["ok"].all()
    ip = api.default_indel_params(1.0)
    gsp = api.default_softclip_params(fresh_object=0)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<4I", codes.numel(), 0, stride, istride))
        f.write(_vec(packed))
        f.write(struct.pack("<Q", 1) + bytes(ip))
        f.write(_vec(iblob))
        f.write(_vec(ioff.astype(np.uint64)))
        f.write(_vec(ilen.astype(np.uint32)))
        f.write(_vec(itasks))
        f.write(struct.pack("<Q", 1) + bytes(gsp))
        f.write(_vec(sblob))
        f.write(_vec(soff.astype(np.uint64)))
        f.write(_vec(slen.astype(np.uint32)))
        f.write(_vec(stasks))
    subprocess.run([BIN, "--rescue", str(fin), str(fout)], check=True, timeout=120)
    raw = np.fromfile(fout, np.uint8)
    ni, ns = len(itasks), len(stasks)
    a = ni * want_i.dtype.itemsize
    b = a + ns * want_s.dtype.itemsize
    got_i = raw[:a].view(want_i.dtype)
    got_s = raw[a:b].view(want_s.dtype)
    c = b + ns * stride
    d = c + len(ct) * want_g.dtype.itemsize
    got_c = raw[b:c].reshape(ns, stride)
    got_g = raw[c:d].view(want_g.dtype)
    got_gc = raw[d:].reshape(len(ct), istride)
    np.testing.assert_array_equal(got_g, want_g)
    for i in range(len(ct)):
        n = want_g["cigar_len"][i] + 1
        assert bytes(got_gc[i, :n]) == bytes(want_gc[i, :n]), i
    np.testing.assert_array_equal(got_i, want_i)
    np.testing.assert_array_equal(got_s, want_s)
    for i in range(ns):
        n = want_s["cigar_len"][i] + 1
        assert bytes(got_c[i, :n]) == bytes(want_c[i, :n]), i


@pytest.mark.gpu
def test_mate_rescue_mirror_matches_oracle(tmp_path, oracle):
    """GpuMateRescue (find_pairs<LevMyers>'s chain per hit, sam_matching.cpp:379-457): addMate per read, hit per task from
    the reference's own quantities - the hit's position, the chromosome table as CRefSeqDesc::GetDescription() holds it -
    one flush; against the oracle on the tasks the reference's own statements produce (tests/test_rescue_chain.py)."""
    from oracle import bindings as ob
    from whisper_amd import api
    (codes, packed), (blob, off, lens), desc, mate, hit_pos, seq_id, events, direct, k = common.rescue_workload(1500, 23, 150)
    tasks = common.rescue_tasks(desc, mate, hit_pos, seq_id, events, direct, k)
    stride = 768
    prm = ob.RescueParams(1, 1)
    rc, want, wext = oracle.rescue_mates(packed, blob, off, lens, tasks, params=prm, ext_stride=stride)
    assert rc == 0 and want["ok"].sum() > 400 and (want["aligned"] != want["ok"]).sum() > 100
    hits = np.zeros(len(tasks), np.dtype([("mate", "<u4"), ("hit_pos", "<u4"), ("thr", "<u4"), ("k", "<u4"), ("events", "<u4"),
                                          ("seq_id", "<i4"), ("direct", "<u4"), ("reserved", "<u4")]))
    hits["mate"], hits["hit_pos"], hits["thr"] = mate, hit_pos, tasks["distance_threshold"]
    hits["k"], hits["events"], hits["seq_id"], hits["direct"] = k, events, seq_id, direct
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<4I", codes.numel(), 0, stride, common.SEPARATE_N_LEN))
        f.write(_vec(packed))
        f.write(struct.pack("<Q", 1) + bytes(api.RescueParams(1, 1)))
        f.write(struct.pack("<Q", 1) + bytes(api.default_scoring()))
        f.write(struct.pack("<Q", len(desc)) + np.ascontiguousarray(desc, np.uint64).tobytes())   # rows = SeqDesc
        f.write(_vec(blob))
        f.write(_vec(off.astype(np.uint64)))
        f.write(_vec(lens.astype(np.uint32)))
        f.write(_vec(hits))
    subprocess.run([BIN, "--mates", str(fin), str(fout)], check=True, timeout=120)
    raw = np.fromfile(fout, np.uint8)
    a = len(tasks) * want.dtype.itemsize
    got = raw[:a].view(want.dtype)
    gext = raw[a:].reshape(len(tasks), stride)
    for fld in want.dtype.names:
        np.testing.assert_array_equal(got[fld], want[fld], err_msg=fld)
    for i in np.nonzero(want["ok"])[0]:
        e = want["ext_cigar_len"][i] + 1
        assert bytes(gext[i, :e]) == bytes(wext[i, :e]), i


def test_integration_patch_compiles():
    """INTEGRATION.md sections 3-4 as an edit of a scratch copy of the reference's mapping_core_sa.cpp,
    mapping_core.cpp and sam_matching.cpp (outside the repository), compiled against gpu_verifier.hpp.
    Needs the reference's sources: skipped where they are absent (the GPU box)."""
    import sys
    if not os.path.isdir("/root/reference/src/mapper"):
        pytest.skip("reference sources absent")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_integration_patch.py")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("compiles") == 7 and "per-bin    mapping_core.cpp       compiles" in r.stdout
    assert "per-bin    mapping_core_dp.cpp    compiles" in r.stdout   # the malicious groups' loop


def _bin_expected(oracle, z, per_cand_k):
    """What the loop body leaves in cur_match for every candidate of a saloop fixture (mapping_core_sa.cpp:223-255,
    279-307), from the oracle's CF pre-filter and verifier on the fixture's tasks."""
    ref_size, cur_offset = int(z["ref_size"]), int(z["seg"][0])
    blob, off, lens = z["blob"], z["off"], z["lens"]
    _, cf = oracle.cf_filter(z["packed"], ref_size, cur_offset, blob, off, lens, z["cf"])
    passed = (z["cf"]["pattern_len"] == 0) | (cf <= 4 * per_cand_k.astype(np.uint32))
    lev_t, dist_t = z["lev"].copy(), z["dist"].copy()
    lev_t["max_mismatches"] = per_cand_k
    dist_t["max_ed"] = per_cand_k
    _, e_lev = oracle.levdiag(z["packed"], ref_size, cur_offset, blob, off, lens, lev_t)
    _, d = oracle.distance(z["packed"], blob, off, lens, 0, dist_t, -1)
    return passed, e_lev, d["edit_distance"]


@pytest.mark.gpu
@pytest.mark.parametrize("name,lev_thr", [("saloop_m150_k5_seg2", 7), ("saloop_m150_k12_seg3_sens", 7),
                                          ("saloop_m100_k2_seg1", 2)])
def test_bin_replay_matches_oracle(tmp_path, oracle, name, lev_thr):
    """replay_batch --bin: a bin of a stage through GpuSaBin - addRead per read, candidate per suffix-array entry, ONE
    flush - on the reads and entries of a fixture that pins the reference's own loops; every read verifies with its
    own max_mismatches, so the bin holds reads of both verifiers (below and from lev_alg_thr on)."""
    import golden_util as gu
    z = gu.load(name)
    k, fl, fr, mrl = int(z["k"]), int(z["seg"][0]), int(z["seg"][1]), int(z["min_read_len"])
    nr, n = len(z["read_ids"]), len(z["cf"])
    rf = z["read_first"].astype(np.int64)
    direct = z["cf"]["dir_gen_flag"].astype(np.int64)
    counts = np.zeros((nr, 2), np.uint16)
    for r in range(nr):
        d = int(direct[rf[r]:rf[r + 1]].sum())
        counts[r] = (d, rf[r + 1] - rf[r] - d)
    rng = np.random.default_rng(8)
    read_k = rng.integers(max(1, k - 3), k + 3, nr).astype(np.uint8)       # around the fixture's k, across lev_alg_thr where it is 7
    per_cand = np.repeat(read_k, counts.sum(axis=1).astype(np.int64))
    passed, e_lev, e_dist = _bin_expected(oracle, z, per_cand)
    e = np.where(per_cand >= lev_thr, e_dist, e_lev)
    want = np.where(passed & (e <= per_cand), e, per_cand.astype(np.uint32) + 1).astype(np.uint8)
    assert (want <= per_cand).sum() > n // 10
    lens = z["lens"][:nr]
    stride = (int(lens.max()) + 1) // 2
    reads = np.zeros((nr, stride), np.uint8)
    for r in range(nr):
        b = (int(lens[r]) + 1) // 2
        reads[r, :b] = z["blob"][int(z["off"][r]):int(z["off"][r]) + b]
    fin, fout = tmp_path / "bin.in", tmp_path / "bin.out"
    with open(fin, "wb") as f:
        f.write(struct.pack("<8I", int(z["ref_size"]), fl, nr, fl, fr, mrl, lev_thr, stride))
        for a in (z["packed"], reads, lens.astype(np.uint32), read_k, counts, z["cf"]["position"].astype(np.uint32)):
            f.write(_vec(a))
    out = subprocess.run([BIN, "--bin", str(fin), str(fout), "2"], check=True, timeout=120, stdout=subprocess.PIPE, text=True)
    assert "flush_candidates_per_s" in out.stdout
    got = np.fromfile(fout, np.uint8)
    np.testing.assert_array_equal(got, want)
