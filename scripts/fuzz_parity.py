"""Randomised differential run of the HIP entry points against the oracle (debug aid / evidence, not a test of the suite).

The parity tests use batches that are uniform in read length, k, window and variant; the kernels take wave-uniform
decisions (tie-break rule, band fit, fixed-phase start, light columns, dword count of the batch) from what the lanes
of a wave hold.  Here every batch mixes those per task - read lengths on either side of every dword / variant border,
k from 0 up, windows shorter than the read up to rescue size, forced variants, both strands, both encodings, both
distance modes - and the shapes change from round to round.  The other families run their seeded workloads of
tests/common.py with parameters drawn per round.

  python scripts/fuzz_parity.py --seconds 120 --seed 1        (GPU box; exit status 1 on the first family that differs)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import common  # noqa: E402
from oracle import bindings as ob  # noqa: E402
from whisper_amd import api, synth  # noqa: E402


def mixed_distance_batch(rng, gen, n_groups, per_group, max_m, enc, rescue_frac=0.1):
    """Tasks over reads of many lengths in one batch, shuffled so that every wave mixes them."""
    import torch
    codes, packed = gen
    G = codes.numel()
    borders = [31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 150, 151, 160, 191, 192, 193, 224, 250, 255, 256, 257, 300, 320]
    reads_codes, tasks = [], []
    for _ in range(n_groups):
        if rng.random() < 0.5:
            m = int(rng.choice([b for b in borders if b <= max_m]))
        else:
            m = int(rng.integers(8, max_m + 1))
        kmax = max(1, min(m // 3, 44))
        k = int(rng.integers(0, kmax + 1))
        rd = synth.make_reads(codes, per_group, m, k, int(rng.integers(1, 1 << 30)),
                              true_frac=float(rng.choice([0.05, 0.25, 0.5, 0.9])))
        p = rd["p"].numpy()
        rcf = rd["is_rc"].numpy().astype(np.uint8)
        kind = rng.random()
        for i in range(per_group):
            pre = k
            W = m + 2 * k
            if kind < rescue_frac:                        # a mate-rescue window around the locus
                pre = k + int(rng.integers(0, 400))
                W = pre + m + k + int(rng.integers(0, 300))
            elif kind < 0.3:                              # ragged windows, task by task
                pre = int(rng.integers(0, 2 * k + 3))
                W = max(1, m + int(rng.integers(-k - 2, 2 * k + 12)))
            elif kind < 0.4:                              # windows shorter than the read
                W = max(1, int(rng.integers(1, m + 1)))
            ref_pos = int(min(max(p[i] - pre, 0), G - W - 1))
            variant = 0
            v = rng.random()
            if v < 0.1:
                variant = 1                                # M64 on any read
            elif v < 0.15 and m <= 256:
                variant = 3                                # M256 below 128 symbols too
            elif v < 0.2 and m <= 128:
                variant = 2
            tasks.append((len(reads_codes), ref_pos, W, k, variant, int(rcf[i])))
            reads_codes.append(rd["codes"][i])
    order = rng.permutation(len(tasks))
    tasks = np.array(tasks, api.DIST_TASK)[order]
    blob, off, lens = common.mixed_reads(reads_codes, enc)
    return packed, (blob, off, lens), tasks


def check_distance(ctx, oracle, rng, gen, report):
    enc = int(rng.integers(0, 2))
    max_m = int(rng.choice([60, 128, 160, 256, 320, 420]))
    packed, (blob, off, lens), tasks = mixed_distance_batch(rng, gen, 40, 96, max_m, enc)
    # (-1 with ASCII reads: the oracle and the library both take a fresh object's 0; the reference's own answer there
    #  depends on the object's previous read, so --cpu-reference passes the 0 explicitly)
    last = int(rng.choice([-1, -1, 0, 2, 3]))
    if enc == 1 and last < 0 and isinstance(ctx, RefCtx):
        last = 0
    rc, want = oracle.distance(packed, blob, off, lens, enc, tasks, last)
    assert rc == 0, rc
    reads = ctx.upload_reads(blob, off, lens, enc)
    got = ctx.dynamic_programming(reads, tasks, last)
    bad = np.nonzero((got["edit_distance"] != want["edit_distance"]) | (got["pos"] != want["pos"]))[0]
    report("distance exact (max m %d, enc %d, last %d)" % (max_m, enc, last), len(tasks), bad, tasks, got, want)
    ctx.set_distance_mode(api.DISTANCE_ACCEPTED_ONLY)
    try:
        got = ctx.dynamic_programming(reads, tasks, last)
    finally:
        ctx.set_distance_mode(api.DISTANCE_EXACT)
    k = tasks["max_ed"]
    acc = want["edit_distance"] <= k
    w2 = want.copy()
    w2["edit_distance"][~acc] = k[~acc] + 1
    w2["pos"][~acc] = 0
    bad = np.nonzero((got["edit_distance"] != w2["edit_distance"]) | (got["pos"] != w2["pos"]))[0]
    report("distance accepted-only (max m %d)" % max_m, len(tasks), bad, tasks, got, w2)
    # the alignment of the same tasks
    refine = int(rng.integers(0, 2))
    stride = 4 * ((2 * (max_m + 44) + 8 + 3) // 4)
    last_a = 0 if last < 0 else last
    rc, wres, wcig = oracle.align(packed, blob, off, lens, enc, tasks, last_a, refine, cigar_stride=stride)
    assert rc == 0, rc
    gres, gcig = ctx.align(reads, tasks, last_a, refine, cigar_stride=stride)
    if gres is None:
        reads.free()
        return
    badm = np.zeros(len(tasks), bool)
    for f in wres.dtype.names:
        badm |= gres[f] != wres[f]
    for i in np.nonzero(wres["ok"])[0]:
        ln = int(wres["cigar_len"][i]) + 1
        if bytes(gcig[i, :ln]) != bytes(wcig[i, :ln]):
            badm[i] = True
    if isinstance(ctx, RefCtx):
        # a read of up to 64 symbols can end in column W + 1 (LevMyers128.hpp:115: j runs to W + 1, word_num = 0);
        # getExtCigar then reads tmp_ref_sequence[W], which its caller never filled (sam_matching.cpp:713 copies W
        # symbols): the reference's answer depends on the call before.  The oracle and the library take the genome's symbol
        badm &= ~((wres["ok"] == 1) & (wres["end_pos"] == tasks["max_distance_in_ref"] + 1))
    report("align (refine %d)" % refine, len(tasks), np.nonzero(badm)[0], tasks, gres, wres)
    reads.free()


def check_levdiag(ctx, oracle, rng, gen, report):
    codes, packed, (blob, off, lens), tasks = common.levdiag_odd_workload(3000, int(rng.integers(1, 1 << 30)))
    rc, want = oracle.levdiag(packed, codes.numel(), 0, blob, off, lens, tasks)
    assert rc == 0
    reads = ctx.upload_reads(blob, off, lens, 0, orientations=1)
    got = ctx.lev_diag(reads, tasks)
    reads.free()
    report("levdiag odd shapes", len(tasks), np.nonzero(got != want)[0], tasks, got, want)
    m = int(rng.choice([40, 75, 100, 150, 151, 200, 250, 256, 300]))
    k = int(rng.integers(1, 7))
    fl, fr = common.stage_segments(m, k)[int(rng.integers(0, k + 1))]
    codes, packed, (blob, off, lens), tasks = common.levdiag_workload(m, k, fl, fr, 4000, int(rng.integers(1, 1 << 30)))
    rc, want = oracle.levdiag(packed, codes.numel(), 0, blob, off, lens, tasks)
    reads = ctx.upload_reads(blob, off, lens, 0, orientations=1)
    got = ctx.lev_diag(reads, tasks)
    reads.free()
    report("levdiag m %d k %d seg %d-%d" % (m, k, fl, fr), len(tasks), np.nonzero(got != want)[0], tasks, got, want)


def check_cf(ctx, oracle, rng, gen, report):
    k = int(rng.integers(1, 9))
    min_len = int(rng.choice([100, 150, 151, 250, 300]))
    cur_offset = int(rng.integers(0, 70))
    codes, packed, (blob, off, lens), tasks = common.cf_workload(4000, seed=int(rng.integers(1, 1 << 30)), k=k,
                                                                 cur_offset=cur_offset, min_len=min_len)
    rc, want = oracle.cf_filter(packed, codes.numel(), cur_offset, blob, off, lens, tasks)
    assert rc == 0
    ctx.set_reference_params(codes.numel(), cur_offset)
    try:
        reads = ctx.upload_reads(blob, off, lens, 0, orientations=1)
        got = ctx.cf_filter(reads, tasks)
        reads.free()
    finally:
        ctx.set_reference_params(codes.numel(), 0)
    report("cf filter k %d min_len %d" % (k, min_len), len(tasks), np.nonzero(got != want)[0], tasks, got, want)


def check_indel(ctx, oracle, rng, gen, report):
    m = int(rng.choice([100, 150, 151, 250, 300]))
    enc = int(rng.integers(0, 2))
    mcf = float(rng.choice([0.5, 0.6, 1.0]))
    max_indel = int(rng.choice([30, 50, 80, 100]))
    stage = int(rng.choice([4, 5, 8]))
    codes, packed, (blob, off, lens), tasks = common.indel_workload(1500, m, seed=int(rng.integers(1, 1 << 30)), encoding=enc,
                                                                   stage_major=stage, max_indel=max_indel,
                                                                   max_mm=12 if m <= 250 else 20)
    _, want = oracle.indel_match(packed, blob, off, lens, enc, tasks, ob.default_indel_params(mcf))
    reads = ctx.upload_reads(blob, off, lens, enc, orientations=1)
    got = ctx.indel_match(reads, tasks, api.default_indel_params(mcf))
    bad = np.nonzero(got != want)[0]
    report("indel match m %d enc %d mcf %.1f max_indel %d" % (m, enc, mcf, max_indel), len(tasks), bad, tasks, got, want)
    ct = common.indel_cigar_tasks(tasks, want)
    stride = 4 * ((2 * (m + 2 * max_indel + 20) + 8 + 3) // 4)
    _, wres, wcig = oracle.indel_ext_cigar(packed, blob, off, lens, enc, ct, cigar_stride=stride)
    gres, gcig = ctx.indel_ext_cigar(reads, ct, cigar_stride=stride)
    reads.free()
    badm = gres != wres
    for i in range(len(wres)):
        n = wres["cigar_len"][i] + 1
        if bytes(gcig[i, :n]) != bytes(wcig[i, :n]):
            badm[i] = True
    report("indel ext-cigar", len(ct), np.nonzero(badm)[0], ct, gres, wres)


def check_softclip(ctx, oracle, rng, gen, report):
    lowcx = int(rng.integers(0, 2))
    fresh = int(rng.integers(0, 2))
    enc = int(rng.integers(0, 2))
    m = int(rng.choice([100, 150, 151, 250]))
    W = int(rng.choice([260, 500, 700, 1427, 2500])) + m - 150
    g2 = common.low_complexity_genome() if lowcx else gen
    codes, packed, (blob, off, lens), tasks = common.softclip_workload(1500, seed=int(rng.integers(1, 1 << 30)), m=m, W=W,
                                                                       encoding=enc, gen=g2, lens_jitter=bool(fresh))
    mql = max(151, m + 1)
    _, want, wcig = oracle.softclip_match(packed, blob, off, lens, enc, tasks,
                                          ob.default_softclip_params(max_query_len=mql, fresh_object=fresh), cigar_stride=768)
    c2 = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    try:
        reads = c2.upload_reads(blob, off, lens, enc, orientations=3)
        got, gcig = c2.softclip_match(reads, tasks, api.default_softclip_params(max_query_len=mql, fresh_object=fresh),
                                      cigar_stride=768)
        reads.free()
    finally:
        c2.destroy()
    badm = got != want
    for i in range(len(want)):
        n = want["cigar_len"][i] + 1
        if bytes(gcig[i, :n]) != bytes(wcig[i, :n]):
            badm[i] = True
    report("softclip m %d W %d lowcx %d fresh %d enc %d" % (m, W, lowcx, fresh, enc), len(tasks), np.nonzero(badm)[0],
           tasks, got, want)


def check_post(ctx, oracle, rng, gen, report):
    m = int(rng.choice([100, 150, 151, 250]))
    stride = int(rng.choice([4 * ((2 * m + 60) // 4), 512, 640, 388 if m <= 150 else 644]))
    n = int(rng.integers(500, 3000))
    name = rng.choice(["default", "asym", "odd"])
    sc_o, sc_a = common.scoring(name), common.scoring(name, api)
    ctx.set_scoring(sc_a or api.default_scoring())
    try:
        t, rec, cig = common.clip_illegal_workload(oracle, n, seed=int(rng.integers(1, 1 << 30)), m=m, scoring=sc_o)
        if cig.shape[1] != stride:
            out = np.zeros((cig.shape[0], stride), np.uint8)
            w = min(stride, cig.shape[1])
            out[:, :w] = cig[:, :w]
            if (cig[:, w:] != 0).any():
                return
            cig = out
        rc, wrec, wcig = oracle.clip_illegal(t, rec, cig, scoring=sc_o)
        assert rc == 0
        grec, gcig = ctx.clip_illegal(t, rec, cig)
        badm = np.zeros(len(wrec), bool)
        for f in wrec.dtype.names:
            if f != "reserved":
                badm |= grec[f] != wrec[f]
        for i in range(len(wrec)):
            k = wrec["cigar_len"][i] + 1
            if bytes(gcig[i, :k]) != bytes(wcig[i, :k]):
                badm[i] = True
        report("clip illegal m %d stride %d %s" % (m, stride, name), n, np.nonzero(badm)[0], t, grec, wrec)
        _, wrec2, wcig2 = oracle.clip_boundary(wrec, wcig, scoring=sc_o)
        grec2, gcig2 = ctx.clip_boundary(wrec, wcig)
        badm = grec2 != wrec2
        for i in range(len(wrec2)):
            k = wrec2["cigar_len"][i] + 1
            if bytes(gcig2[i, :k]) != bytes(wcig2[i, :k]):
                badm[i] = True
        report("clip boundary", n, np.nonzero(badm)[0], wrec, grec2, wrec2)
        rc, wc, wm, wb, wl = oracle.sam_cigar(wcig2)
        gc, gm, gb, gl = ctx.sam_cigar(wcig2)
        badm = gl != wl
        for i in range(len(wl)):
            if bytes(gc[i, :wl["cigar_len"][i] + 1]) != bytes(wc[i, :wl["cigar_len"][i] + 1]) or \
               bytes(gm[i, :wl["mdz_len"][i] + 1]) != bytes(wm[i, :wl["mdz_len"][i] + 1]):
                badm[i] = True
        report("sam cigar", n, np.nonzero(badm)[0], wl, gl, wl)
        tp, recp, cigp = common.clip_pairs_workload(oracle, n // 2, seed=int(rng.integers(1, 1 << 30)), m=m, scoring=sc_o)
        rc, wrecp, wcigp = oracle.clip_overlapping_pairs(tp, recp, cigp)
        assert rc == 0
        grecp, gcigp = ctx.clip_overlapping_pairs(tp, recp, cigp)
        badm = np.zeros(len(wrecp), bool)
        for f in wrecp.dtype.names:
            if f != "reserved":
                badm |= grecp[f] != wrecp[f]
        for i in range(len(wrecp)):
            k = wrecp["cigar_len"][i] + 1
            if bytes(gcigp[i, :k]) != bytes(wcigp[i, :k]):
                badm[i] = True
        report("clip pairs", len(wrecp), np.nonzero(badm)[0], recp, grecp, wrecp)
    finally:
        ctx.set_scoring(api.default_scoring())


def check_finish(ctx, oracle, rng, gen, report):
    m = int(rng.choice([100, 150, 151, 250]))
    flank = int(rng.integers(4, 15))
    fpf = float(rng.choice([0.0, 0.7]))
    name = rng.choice(["default", "asym", "odd"])
    sc_o, sc_a = common.scoring(name), common.scoring(name, api)
    ctx.set_scoring(sc_a or api.default_scoring())
    try:
        (codes, packed), (blob, off, lens), desc, ridx, maps, direct, distant, stale = common.finish_workload(
            oracle, 1000, int(rng.integers(1, 1 << 30)), m=m, flank=flank, scoring=sc_o, from_pair_frac=fpf)
        kw = dict(isize=common.FINISH_ISIZE) if fpf else {}
        tasks = common.finish_tasks(desc, ridx, lens, maps, direct, distant, stale, scoring=sc_o, **kw)
        rc, *want = oracle.finish_pairs(packed, blob, off, lens, tasks, ob.default_finish_params(flank), scoring=sc_o,
                                        ext_stride=768)
        assert rc == 0
        reads = ctx.upload_reads(blob, off, lens, 1, orientations=3)
        try:
            got = ctx.finish_pairs(reads, tasks, api.default_finish_params(flank), ext_stride=768)
        except api.WamdError:
            # a record whose clipping runs past the read: the reference's unsigned lengths wrap (the oracle restates
            # that: err_edit_distance 65533, a left clipping longer than the read), the library refuses the batch
            #  - about one generated slot in a million
            w = want[0]
            wrapped = (w["err_edit_distance"] >= 0x8000) | (w["left_clipping"].astype(np.int64) + w["right_clipping"] > m + 1)
            reads.free()
            if not wrapped.any():
                raise
            print("   finish: batch refused for %d wrapping record(s), as documented" % int(wrapped.sum()), flush=True)
            report("finish pairs refused (wrapping record)", len(tasks), np.zeros(0, int), tasks, want[0], want[0])
            return
        reads.free()
        try:
            common.same_finish_output(got, want)
            bad = np.zeros(0, int)
        except AssertionError as e:
            print(str(e)[:800])
            bad = np.arange(1)
        report("finish pairs m %d flank %d %s from_pair %.1f" % (m, flank, name, fpf), len(tasks), bad, tasks, got[0], want[0])
    finally:
        ctx.set_scoring(api.default_scoring())


def check_rescue(ctx, oracle, rng, gen, report):
    m = int(rng.choice([100, 150, 151, 250]))
    name = rng.choice(["default", "asym", "odd"])
    sc_o = common.scoring(name) or ob.default_scoring()
    sc_a = common.scoring(name, api) or api.default_scoring()
    (codes, packed), (blob, off, lens), desc, mate, hit_pos, seq_id, events, direct, k = common.rescue_workload(
        2000, int(rng.integers(1, 1 << 30)), m)
    tasks = common.rescue_tasks(desc, mate, hit_pos, seq_id, events, direct, k)
    rc, want, wext = oracle.rescue_mates(packed, blob, off, lens, tasks, scoring=sc_o, ext_stride=768)
    assert rc == 0
    c2 = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    try:
        c2.set_scoring(sc_a)
        reads = c2.upload_reads(blob, off, lens, api.READS_ASCII, orientations=3)
        got, gext = c2.rescue_mates(reads, tasks, ext_stride=768)
        reads.free()
    finally:
        c2.destroy()
    badm = np.zeros(len(want), bool)
    for f in want.dtype.names:
        badm |= got[f] != want[f]
    for i in np.nonzero(want["ok"])[0]:
        e = want["ext_cigar_len"][i] + 1
        if bytes(gext[i, :e]) != bytes(wext[i, :e]):
            badm[i] = True
    report("rescue mates m %d %s" % (m, name), len(tasks), np.nonzero(badm)[0], tasks, got, want)


def check_groups(ctx, oracle, rng, gen, report):
    """The commit block of the mapping loop and the bin's result-group bytes (rows K / (f)3)."""
    n_reads = int(rng.integers(200, 6000))
    k = int(rng.integers(1, 17))
    max_cand = int(rng.integers(1, 40))
    mnm = int(rng.choice([1, 2, 3, 8, 60, 1024]))
    id_bytes = int(rng.integers(3, 9))
    t, indel, read_first, read_ids = common.commit_workload(oracle, n_reads, int(rng.integers(1, 1 << 30)), k=k, max_cand=max_cand)
    if id_bytes < 8:
        read_ids = read_ids & np.uint64((1 << (8 * id_bytes)) - 1)
    _, want_c = oracle.commit(t, indel)
    rc, want = oracle.result_groups(t, want_c, read_first, read_ids, ob.GroupParams(id_bytes, 4, mnm, 0))
    assert rc == 0
    committed, data, offs, flags, _ = common.gpu_result_groups(ctx, t, indel, read_first, read_ids, id_bytes, mnm)
    bad = np.nonzero(committed != want_c)[0]
    if not len(bad) and (flags or len(data) != len(want) or bytes(data) != bytes(want)):
        print("   group bytes differ: flags %d, %d bytes against %d" % (flags, len(data), len(want)))
        bad = np.arange(1)
    report("groups reads %d k %d max_cand %d max_no_mappings %d id_bytes %d" % (n_reads, k, max_cand, mnm, id_bytes),
           len(t), bad, t, committed, want_c)


def saloop_case(rng, ref):
    """A fresh workload of the suffix-array loops, run through the reference's own loops (as tests/golden/make_golden.py
    makes the saloop_* fixtures): the dictionary tests/test_gpu_sa_loop.py reads."""
    m = int(rng.choice([100, 150, 151, 250]))
    sensitive = int(rng.integers(0, 2))
    k = int(rng.integers(8, 17)) if sensitive else int(rng.integers(1, 13))
    stage = int(k / 2.5) if sensitive else k
    segi = int(rng.integers(0, stage + 1))
    seg = common.stage_segments(m, stage)[segi]
    codes, packed, reads, sa_dir, sa_rc = common.sa_loop_workload(int(rng.integers(1, 1 << 30)), 200, 3, m, k, seg, min_read_len=m)
    ref_size = codes.numel()
    prm = ob.default_indel_params(0.6)
    rc, ranges, best, pushed, pdir, pread = ref.search_reads(packed, ref_size, *reads, sa_dir, sa_rc, seg, m, k,
                                                             sensitive=sensitive, enable_indels=sensitive, params=prm)
    assert rc == 0, rc
    n = len(reads[2])
    reads2 = common.with_reverse_reads(reads)
    cf, dist, lev, indel, commit, read_first = common.sa_loop_tasks(ranges, reads[2], sa_dir, sa_rc, ref_size, seg, m, k, 50)
    committed = np.zeros(len(commit), ob.MAPPING)
    cand_read = np.repeat(np.arange(n), np.diff(read_first).astype(np.int64))
    j = 0
    for i in range(len(commit)):
        if j < len(pushed) and pread[j] == cand_read[i] and pdir[j] == commit["direct"][i] and (
                int(pushed[j]["pos"]) == int(commit[i]["lev_pos"]) if pushed[j]["type"] == 0
                else abs(int(pushed[j]["pos"]) - int(commit[i]["lev_pos"])) <= m + 64):
            committed[i] = pushed[j]
            committed[i]["found"] = 1
            j += 1
        else:
            committed[i]["type"] = 8
    assert j == len(pushed), (j, len(pushed))
    read_ids = (np.arange(n, dtype=np.uint64) * 3 + (5 << 32)).astype(np.uint64)
    commit_full = commit.copy()
    commit_full["has_indel"] = sensitive
    rc, group = ref.result_groups(commit_full, committed, read_first, read_ids, ob.GroupParams(5, 4, 1024, 0))
    assert rc == 0
    blob, off, lens = reads2
    z = dict(packed=packed, ref_size=np.uint32(ref_size), blob=blob, off=off, lens=lens, n_reads=n,
             seg=np.array(seg, np.uint32), k=k, min_read_len=m, sensitive=sensitive, min_clipped_factor=0.6, cf=cf, dist=dist,
             lev=lev, indel=indel, commit=commit, read_first=read_first, read_ids=read_ids, pushed=pushed, pushed_direct=pdir,
             group_bytes=group)
    return "m %d k %d seg %d sensitive %d (%d candidates, %d pushed)" % (m, k, segi, sensitive, len(commit), len(pushed)), z


def check_saloop(ctx, oracle, rng, gen, report):
    """The device leg of the mapping loop (filter + verify -> Match -> commit -> result groups, and the compact form)
    against the reference's search_in_*_range_dir_and_rc on a fresh workload; needs oracle/_ref."""
    if not ob.have_reference():
        return
    label, z = saloop_case(rng, ob.Reference())
    if isinstance(ctx, RefCtx):     # the CPU run stops at "the reference's loops ran and their pushes are accounted for"
        report("saloop " + label, len(z["commit"]), np.zeros(0, int), None, None, None)
        return
    import golden_util
    import test_gpu_sa_loop as T
    keep = golden_util.load
    golden_util.load = lambda name: z
    bad = np.zeros(0, int)
    try:
        for mode in (api.DISTANCE_EXACT, api.DISTANCE_ACCEPTED_ONLY):
            T.test_device_loop_equals_the_reference_loop("fuzz", mode)
        for filtered in (True, False):
            T.test_compact_form_equals_the_task_form("fuzz", filtered)
    except AssertionError:
        import traceback
        print(traceback.format_exc()[-1500:], flush=True)
        bad = np.arange(1)
    finally:
        golden_util.load = keep
    report("saloop " + label, len(z["commit"]), bad, [label], ["see above"], [""])


class RefCtx:
    """The reference's own LevMyers objects behind the calls check_distance makes (--cpu-reference)."""

    def __init__(self, gen):
        self.ref = ob.Reference()
        self.codes, self.packed = gen
        self.mode = api.DISTANCE_EXACT

    class _Reads:
        def free(self):
            pass

    def upload_reads(self, blob, off, lens, enc, orientations=3):
        r = RefCtx._Reads()
        r.blob, r.off, r.lens, r.enc = blob, off, lens, enc
        return r

    def set_distance_mode(self, mode):
        self.mode = mode

    def dynamic_programming(self, r, tasks, last):
        rc, res = self.ref.distance(self.packed, self.codes.numel(), r.blob, r.off, r.lens, r.enc, tasks, last, nthreads=8)
        assert rc == 0, rc
        if self.mode == api.DISTANCE_ACCEPTED_ONLY:
            rej = res["edit_distance"] > tasks["max_ed"]
            res["edit_distance"][rej] = tasks["max_ed"][rej] + 1
            res["pos"][rej] = 0
        return res

    def align(self, r, tasks, last, refine, cigar_stride):
        rc, res, cig = self.ref.align(self.packed, self.codes.numel(), r.blob, r.off, r.lens, r.enc, tasks, last, refine,
                                      cigar_stride=cigar_stride, nthreads=8)
        if rc:      # the shim gives up on a batch holding a task the reference's own caller would not make
            print("   (reference align rc %d: alignment of this batch not compared)" % rc)
            return None, None
        return res, cig

    def take_flags(self):
        return 0

    def destroy(self):
        pass


FAMILIES = {"distance": check_distance, "levdiag": check_levdiag, "cf": check_cf, "indel": check_indel,
            "softclip": check_softclip, "post": check_post, "finish": check_finish, "rescue": check_rescue, "groups": check_groups,
            "saloop": check_saloop}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--families", default=",".join(FAMILIES))
    ap.add_argument("--replay", default="", help="family:seed[,family:seed...] as printed with a mismatch")
    ap.add_argument("--cpu-reference", action="store_true",
                    help="no GPU: the compiled reference (oracle/_ref) stands in for the device on the distance family - "
                         "tells an oracle bug from a kernel bug when a batch differs")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    oracle = ob.Oracle()
    gen = common.genome()
    if args.cpu_reference:
        ctx = RefCtx(gen)
        args.families = ",".join(f for f in args.families.split(",") if f in ("distance", "saloop"))
    else:
        ctx = api.Context(gen[1], ref_size=gen[0].numel(), cur_offset=0)
    totals, failed = {}, []
    cur = [""]

    def report(label, n, bad, tasks, got, want):
        fam = label.split(" ")[0]
        t = totals.setdefault(fam, [0, 0, 0])
        t[0] += 1
        t[1] += n
        t[2] += len(bad)
        if len(bad):
            failed.append(label)
            print("MISMATCH [%s] %s: %d of %d" % (cur[0], label, len(bad), n), flush=True)
            for i in bad[:5]:
                print("   task", i, tasks[i] if i < len(tasks) else "", "\n   got ", got[i] if i < len(got) else "",
                      "\n   want", want[i] if i < len(want) else "", flush=True)

    fams = [f for f in args.families.split(",") if f]
    t0 = time.time()
    rnd = 0
    replay = [x.split(":") for x in args.replay.split(",") if x]
    while time.time() - t0 < args.seconds and not failed:
        # every batch has a seed of its own, printed with a mismatch: --replay family:seed runs that batch alone
        todo = replay if replay else [(f, int(rng.integers(1, 1 << 30))) for f in fams]
        for f, bseed in todo:
            cur[0] = "%s:%d" % (f, int(bseed))
            try:
                FAMILIES[f](ctx, oracle, np.random.default_rng(int(bseed)), gen, report)
            except BaseException:
                print("EXCEPTION in batch [%s] (--replay %s)" % (cur[0], cur[0]), flush=True)
                raise
            flags = ctx.take_flags()
            if flags:
                print("device flags 0x%x after %s" % (flags, f), flush=True)
                failed.append("flags " + f)
        rnd += 1
        if replay:
            break
        print("round %d, %.0f s: %s" % (rnd, time.time() - t0,
                                         ", ".join("%s %d batches / %d units" % (k, v[0], v[1]) for k, v in totals.items())),
              flush=True)
    ctx.destroy()
    print("FUZZ", "FAILED: " + "; ".join(failed) if failed else "ok", "seed", args.seed, "rounds", rnd)
    raise SystemExit(1 if failed else 0)


if __name__ == "__main__":
    main()
