"""Generates tests/golden/*.npz: inputs and the outputs of the REFERENCE's own verifier
objects (oracle/_ref/libwhisper_ref.so, built by oracle/Makefile from /root/reference/src)
on small seeded workloads.  Only data is stored - no reference source or binary.

    python tests/golden/make_golden.py        (needs /root/reference; run in the build container)

The fixtures pin oracle/wv_oracle.c (tests/test_oracle_golden.py) and the HIP path
(tests/test_gpu_golden.py) on machines that never see the reference.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import common  # noqa: E402
from oracle import bindings  # noqa: E402
from whisper_amd import synth  # noqa: E402

GENOME_SYMBOLS = 1 << 16
GENOME_SEED = 20211219

# name, m, k, W, variant, encoding, last_symbol, n
DISTANCE = [
    ("dist_m150_k12_nibble", 150, 12, None, 0, 0, -1, 1500),
    ("dist_m150_k12_ascii_last0", 150, 12, None, 0, 1, 0, 1500),
    ("dist_m150_k12_ascii_last3", 150, 12, None, 0, 1, 3, 1000),
    ("dist_m250_k17", 250, 17, None, 0, 0, -1, 1000),
    ("dist_m100_k10", 100, 10, None, 0, 0, -1, 1000),
    ("dist_m300_k20", 300, 20, None, 0, 0, -1, 600),
    ("dist_m150_k26_w660", 150, 26, 660, 0, 1, 0, 600),
    ("dist_m64_k6", 64, 6, None, 0, 0, -1, 600),
    ("dist_m128_k8", 128, 8, None, 0, 0, -1, 600),
    ("dist_m256_k10", 256, 10, None, 0, 0, -1, 600),
    ("dist_m150_k12_m64forced", 150, 12, None, 1, 0, -1, 600),
]
# name, m, k, segment index, n
LEVDIAG = [("lev_m150_k%d_seg%d" % (k, j), 150, k, j, 600)
           for k in (1, 2, 5, 6) for j in range(k + 1)]
# name, m, k, W, last_symbol, refine, n
ALIGN = [
    ("align_m150_k12", 150, 12, None, 0, 1, 1500),
    ("align_m150_k12_norefine", 150, 12, None, 0, 0, 600),
    ("align_m150_k13_w177", 150, 13, 177, 0, 1, 1000),
    ("align_m150_k26_w660", 150, 26, 660, 0, 1, 600),
    ("align_m250_k17", 250, 17, None, 0, 1, 600),
    ("align_m250_k44_w700", 250, 44, 700, 0, 1, 400),
    ("align_m100_k10", 100, 10, None, 0, 1, 600),
    ("align_m300_k20", 300, 20, None, 0, 1, 400),
]


def make_cf(ref, codes, packed, ref_size):
    """CF pre-filter fixtures: the reference's own calculate_CF_128b_* + horizontal_add(abs())."""
    for name, k, cur_offset, n in (("cf_m150_k5", 5, 0, 1500), ("cf_m150_k2_off31", 2, 31, 800)):
        _, _, (blob, off, lens), tasks = common.cf_workload(n, seed=700 + k, k=k, cur_offset=cur_offset,
                                                            gen=(codes, packed))
        rc, res, ok = ref.cf_filter(packed, ref_size, cur_offset, blob, off, lens, tasks, max_diff=4 * k)
        assert rc == 0
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="cf", blob=blob, off=off, lens=lens, tasks=tasks,
                            results=res, passed=ok, cur_offset=cur_offset, max_diff=4 * k)
        print(name, "passed", int(ok.sum()), "of", n)


def make_sam_cigar(ref):
    """Row X fixtures: the reference's extract_cigar_and_mdz / extract_cigarbin_and_mdz."""
    ext = common.ext_cigar_pool(bindings.Oracle(), n=1500, seed=31)
    keep = np.ones(len(ext), bool)
    rc, cig, mdz, cbin, lens = ref.sam_cigar(ext)
    assert rc == 0
    w = int(lens["cigar_len"].max()) + 1, int(lens["mdz_len"].max()) + 1, int(lens["cigar_bin_len"].max())
    np.savez_compressed(os.path.join(HERE, "samcigar_pool.npz"), kind="samcigar", ext=ext[keep][:, :384],
                        cigars=cig[:, :w[0]], mdz=mdz[:, :w[1]], cigar_bin=cbin[:, :w[2]], lens=lens)
    print("samcigar_pool", len(ext), "ext-CIGARs, longest CIGAR", w[0] - 1, "MD:Z", w[1] - 1)


def make_clip_l(ref):
    """Row L fixtures: the reference's clipIllegalPositions and clipOverlappingPairs."""
    orc = bindings.Oracle()
    t, rec, cig = common.clip_illegal_workload(orc, 900, seed=61)
    rc, rec2, cig2 = ref.clip_illegal(t, rec, cig)
    assert rc == 0
    np.savez_compressed(os.path.join(HERE, "clipillegal_m150.npz"), kind="clipillegal", tasks=t, records=rec,
                        cigars=cig, out_records=rec2, out_cigars=cig2)
    print("clipillegal_m150 clipped", int((rec2["no_clippings"] > 0).sum()), "of", len(rec))
    t, rec, cig = common.clip_pairs_workload(orc, 600, seed=62)
    rc, rec2, cig2 = ref.clip_overlapping_pairs(t, rec, cig)
    assert rc == 0
    np.savez_compressed(os.path.join(HERE, "clippairs_m150.npz"), kind="clippairs", tasks=t, records=rec,
                        cigars=cig, out_records=rec2, out_cigars=cig2)
    print("clippairs_m150 changed", int(((rec2["left_clipping"] != rec["left_clipping"]) |
                                         (rec2["right_clipping"] != rec["right_clipping"])).sum()), "of", len(rec))


def make_softclip(ref, codes, packed):
    """Row S fixtures: the reference's CSoftClipping::{preprocessRawSeq, match, getExtCigar}."""
    for name, W, fresh, n in (("softclip_m150_w700", 700, 0, 700), ("softclip_m150_w1427_fresh", 1427, 1, 500)):
        _, _, (blob, off, lens), tasks = common.softclip_workload(n, seed=900 + W, W=W, encoding=1,
                                                                  gen=(codes, packed), lens_jitter=bool(fresh))
        prm = bindings.default_softclip_params(fresh_object=fresh)
        rc, res, cig = ref.softclip_match(packed, blob, off, lens, 1, tasks, prm, cigar_stride=384)
        assert rc == 0
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="softclip", blob=blob, off=off, lens=lens,
                            encoding=1, tasks=tasks, results=res, cigars=cig, cigar_stride=384, fresh_object=fresh,
                            max_query_len=151, max_mismatches=12)
        print(name, "found", int(res["found"].sum()), "of", n, "indels", int((res["del_size"] != 0).sum()))


def make_finish(ref, codes, packed, ref_size):
    """The chosen pair from its two wire records to the SAM fields: the reference's own deserialize, Translate,
    mapping_desc_t, fill_cigar, clipIllegalPositions, clipBoundaryPositions, clipOverlappingPairs and
    extract_cigar_and_mdz in the order of sam.cpp:1555-1580, 1871-1900, 1992-2007 (oracle/ref_shim.cpp,
    ref_finish_pairs_batch).  `tasks` is what the GPU entry point takes: the descriptors as constructed."""
    orc = bindings.Oracle()
    for name, n_pairs, seed, m in (("finish_m150", 400, 8100, 150), ("finish_m100", 150, 8101, 100)):
        (_, _), (blob, off, lens), desc, ridx, maps, direct, distant, stale = common.finish_workload(
            orc, n_pairs, seed, m=m, gen=(codes, packed))
        _, wire, sizes = ref.serialize(maps, direct)
        rc, tasks, res, ext, cig, mdz = ref.finish_pairs(packed, ref_size, desc, blob, off, lens, ridx, wire, distant,
                                                         stale, ext_stride=512)
        assert rc == 0 and res["ok"].all()
        w = [int(res[f].max()) + 1 for f in ("ext_cigar_len", "cigar_len", "mdz_len")]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="finish", blob=blob, off=off, lens=lens, desc=desc,
                            read_idx=ridx, wire=wire, wire_sizes=sizes, distant=distant, stale_penalty=stale,
                            tasks=tasks, results=res, ext_cigars=ext[:, :w[0]], cigars=cig[:, :w[1]], mdz=mdz[:, :w[2]],
                            flank=6, ext_stride=512)
        print(name, "methods", np.bincount(res["method"], minlength=27), "clipped",
              int(((res["left_clipping"] > 0) | (res["right_clipping"] > 0)).sum()))
    # round 3: isDistant computed inside the chain for 60 % of the pairs (WAMD_DISTANT_FROM_PAIR; a model of 330 +- 7 * 12
    # makes part of them distant) and every ninth lev record claiming fewer errors than it has (fill_cigar_with_lev fails:
    # the slot is reported as not finished, the mate is finished)
    (_, _), (blob, off, lens), desc, ridx, maps, direct, distant, stale = common.finish_workload(
        orc, 300, 8102, m=150, gen=(codes, packed), from_pair_frac=0.6)
    lev = np.nonzero((maps["type"] == 0) & (maps["no_mismatches"] > 2))[0]
    maps["no_mismatches"][lev[::9]] = 1
    _, wire, sizes = ref.serialize(maps, direct)
    rc, tasks, res, ext, cig, mdz = ref.finish_pairs(packed, ref_size, desc, blob, off, lens, ridx, wire, distant, stale,
                                                     ext_stride=512, model=common.FINISH_MODEL)
    assert rc == 0 and 5 < (res["ok"] == 0).sum() < 100
    w = [int(res[f].max()) + 1 for f in ("ext_cigar_len", "cigar_len", "mdz_len")]
    np.savez_compressed(os.path.join(HERE, "finish_m150_frompair.npz"), kind="finish", blob=blob, off=off, lens=lens,
                        desc=desc, read_idx=ridx, wire=wire, wire_sizes=sizes, distant=distant, stale_penalty=stale,
                        tasks=tasks, results=res, ext_cigars=ext[:, :w[0]], cigars=cig[:, :w[1]], mdz=mdz[:, :w[2]],
                        flank=6, ext_stride=512)
    print("finish_m150_frompair methods", np.bincount(res["method"], minlength=27), "not finished", int((res["ok"] == 0).sum()))


def make_rescue(ref, codes, packed, ref_size):
    """find_pairs<LevMyers>'s chain per hit (sam_matching.cpp:379-457): the reference's own statements on its own
    objects (oracle/ref_shim.cpp: ref_rescue_mates_batch) - preprocessRawSeq, dynamicProgramming, getExtCigar,
    refine_ext_cigar, Translate, the mate's mapping_desc_t, clipIllegalPositions, clipBoundaryPositions.  `tasks` is
    what the GPU entry point takes."""
    for name, n, seed, m, scn in (("rescue_m150", 600, 8300, 150, None), ("rescue_m100_sc_odd", 300, 8301, 100, "odd")):
        sc = common.scoring(scn) if scn else bindings.default_scoring()
        (_, _), (blob, off, lens), desc, mate, hit_pos, seq_id, events, direct, k = common.rescue_workload(
            n, seed, m, gen=(codes, packed))
        rc, tasks, res, ext = ref.rescue_mates(packed, ref_size, desc, blob, off, lens, mate, hit_pos, seq_id, events, direct, k,
                                               scoring=sc, model=common.RESCUE_MODEL, ext_stride=512)
        assert rc == 0
        w = (int(res["ext_cigar_len"].max()) + 4) // 4 * 4
        extra = {} if scn is None else {"scoring": np.array(common.SCORING_VALUES[scn])}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="rescue", packed=packed, ref_size=np.uint32(ref_size),
                            blob=blob, off=off, lens=lens, desc=desc, tasks=tasks, results=res, ext=ext[:, :w], **extra)
        print(name, "ok", int(res["ok"].sum()), "aligned", int(res["aligned"].sum()), "clipped",
              int(((res["left_clipping"] > 0) | (res["right_clipping"] > 0)).sum()))


def make_groups(ref):
    """Verified candidates -> committed mappings -> result-group bytes, by the reference's construct<lev>, better,
    CMappingsHeapGatherer and CMappingResultsCollector::Push (oracle/ref_shim.cpp: ref_commit_batch,
    ref_result_groups_batch)."""
    orc = bindings.Oracle()
    for name, seed, max_no, idb in (("groups_heap1024_id5", 8200, 1024, 5), ("groups_heap8_id4", 8201, 8, 4)):
        t, indel, read_first, read_ids = common.commit_workload(orc, 1200, seed)
        rc, committed = ref.commit(t, indel)
        assert rc == 0
        rc, data = ref.result_groups(t, committed, read_first, read_ids, bindings.GroupParams(idb, 4, max_no, 0))
        assert rc == 0
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="groups", tasks=t, indel=indel, read_first=read_first,
                            read_ids=read_ids, id_bytes=idb, mapping_counter_size=4, max_no_mappings=max_no,
                            committed=committed, group_bytes=data)
        print(name, "kept", int(committed["found"].sum()), "of", len(t), "group bytes", len(data))


def make_round2(ref, codes, packed, ref_size):
    """Shapes no round-1 fixture reached: LevDiag beyond 160 symbols (BASELINE configs[4] runs it in
    stages 1-4 at 250 bp; 300 / 600 bp for the two widest kernel instantiations), the long-indel
    matcher on reads above 256 symbols and with max_indel_size above 80, the untrained-model
    rescue window W = 1427 (sam.cpp:185-188), the CF pre-filter at min_read_len = 250 and the 250 bp
    k = 7 stage.  Own seeds, so the older fixtures stay byte for byte what they were."""
    seed = 5000
    lev = [("lev_m250_k%d_seg%d" % (k, j), 250, k, j, 300) for k in (1, 2, 3, 4) for j in range(k + 1)]
    lev += [("lev_m300_k4_seg1", 300, 4, 1, 300), ("lev_m300_k6_seg6", 300, 6, 6, 300),
            ("lev_m600_k6_seg0", 600, 6, 0, 200), ("lev_m600_k5_seg3", 600, 5, 3, 200)]
    for name, m, k, j, n in lev:
        seed += 1
        fl, fr = common.stage_segments(m, k)[j]
        rd = synth.make_reads(codes, n, m, k, seed, rc_frac=0.0, protect=(fl, fr))
        blob, off, lens = common.host_reads(rd, 0)
        tasks = synth.levdiag_tasks(rd, k, fl, fr, 0)
        rc, res = ref.levdiag(packed, ref_size, 0, blob, off, lens, tasks)
        assert rc == 0
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="levdiag", blob=blob, off=off,
                            lens=lens, tasks=tasks, results=res, cur_offset=0)
        print(name, "accepted", int((res <= k).sum()), "of", n)
    for name, m, max_indel, max_mm, stage, mcf, n in [("indel_m300_maxindel50", 300, 50, 20, 8, 1.0, 500),
                                                      ("indel_m150_maxindel100", 150, 100, 12, 5, 0.6, 500)]:
        seed += 1
        _, _, (blob, off, lens), tasks = common.indel_workload(n, m, seed, max_indel=max_indel, max_mm=max_mm,
                                                               stage_major=stage, gen=(codes, packed))
        prm = bindings.default_indel_params(mcf)
        rc, res = ref.indel_match(packed, blob, off, lens, 0, tasks, prm)
        ct = common.indel_cigar_tasks(tasks, res)
        stride = 4 * ((2 * (m + 2 * max_indel) + 16 + 3) // 4)
        rc, cres, ccig = ref.indel_ext_cigar(packed, blob, off, lens, 0, ct, max_indel=max_indel, cigar_stride=stride)
        direction = (np.arange(len(res)) % 3 != 0).astype(np.uint8)
        rc, wire, sizes = ref.serialize(res, direction)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="indel", blob=blob, off=off, lens=lens,
                            encoding=0, min_clipped_factor=mcf, tasks=tasks, results=res,
                            cigar_tasks=ct, cigar_results=cres, cigars=ccig, cigar_stride=stride,
                            direction=direction, wire=wire, wire_sizes=sizes)
        print(name, "found", int(res["found"].sum()), "types", np.bincount(res["type"][res["found"] == 1], minlength=9))
    for name, m, k, W, n in [("align_m150_k26_w1427", 150, 26, 1427, 400), ("align_m250_k44_w1500", 250, 44, 1500, 200)]:
        seed += 1
        rd = synth.make_reads(codes, n, m, k, seed)
        blob, off, lens = common.host_reads(rd, 1)
        tasks = synth.distance_tasks(rd, m, k, W, 0)
        stride = 4 * ((2 * (m + k) + 8 + 3) // 4)
        rc, res, cig = ref.align(packed, ref_size, blob, off, lens, 1, tasks, 0, 1, cigar_stride=stride)
        assert rc == 0
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="align", blob=blob, off=off,
                            lens=lens, encoding=1, last_symbol=0, refine=1, tasks=tasks,
                            results=res, cigars=cig, cigar_stride=stride)
        print(name, "ok", int(res["ok"].sum()), "refined", int(res["refined"].sum()), "of", n)
    for name, k, cur_offset, n in (("cf_m250_k7", 7, 0, 800), ("cf_m250_k3_off62", 3, 62, 600)):
        _, _, (blob, off, lens), tasks = common.cf_workload(n, seed=5700 + k, min_len=250, k=k, cur_offset=cur_offset,
                                                            gen=(codes, packed))
        rc, res, ok = ref.cf_filter(packed, ref_size, cur_offset, blob, off, lens, tasks, max_diff=4 * k)
        assert rc == 0
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="cf", blob=blob, off=off, lens=lens, tasks=tasks,
                            results=res, passed=ok, cur_offset=cur_offset, max_diff=4 * k)
        print(name, "passed", int(ok.sum()), "of", n)
    make_finish(ref, codes, packed, ref_size)
    make_groups(ref)
    seed += 1
    rd = synth.make_reads(codes, 600, 250, 7, seed)
    blob, off, lens = common.host_reads(rd, 0)
    tasks = synth.distance_tasks(rd, 250, 7, None, 0)
    rc, res = ref.distance(packed, ref_size, blob, off, lens, 0, tasks, -1)
    assert rc == 0
    np.savez_compressed(os.path.join(HERE, "dist_m250_k7.npz"), kind="distance", blob=blob, off=off,
                        lens=lens, encoding=0, last_symbol=-1, tasks=tasks, results=res)
    print("dist_m250_k7 accepted", int((res["edit_distance"] <= 7).sum()), "of", 600)


def make_scoring(ref, codes, packed, ref_size):
    """Round 3: every routine that reads a score, under two non-default scoring_t (common.SCORING_VALUES: one
    asymmetric in insertion / deletion costs, one whose values are not multiples of 0.2), from the reference's own
    objects: getExtCigar + refine_ext_cigar, Match + GetExtCigarNew, clipBoundaryPositions, clipIllegalPositions,
    CSoftClipping::getExtCigar and the finish chain (mapping_desc_t's indel_to_mismatch_score included)."""
    orc = bindings.Oracle()
    seed = 9000
    for sc_name in ("asym", "odd"):
        sc = common.scoring(sc_name)
        scv = np.array(common.SCORING_VALUES[sc_name], np.float64)
        seed += 10
        for name, m, k, W, n in (("align_m150_k13_w177", 150, 13, 177, 600), ("align_m150_k26_w660", 150, 26, 660, 300)):
            rd = synth.make_reads(codes, n, m, k, seed + (W % 7))
            blob, off, lens = common.host_reads(rd, 1)
            tasks = synth.distance_tasks(rd, m, k, W, 0)
            stride = 4 * ((2 * (m + k) + 8 + 3) // 4)
            rc, res, cig = ref.align(packed, ref_size, blob, off, lens, 1, tasks, 0, 1, scoring=sc, cigar_stride=stride)
            assert rc == 0
            np.savez_compressed(os.path.join(HERE, "%s_sc_%s.npz" % (name, sc_name)), kind="align", blob=blob, off=off,
                                lens=lens, encoding=1, last_symbol=0, refine=1, tasks=tasks, results=res, cigars=cig,
                                cigar_stride=stride, scoring=scv)
            print(name, sc_name, "ok", int(res["ok"].sum()), "refined", int(res["refined"].sum()), "of", n)
        _, _, (blob, off, lens), tasks = common.planted_refine_workload(200, 150, 12, seed + 1, gen=(codes, packed))
        stride = 4 * ((2 * (150 + 12) + 8 + 3) // 4)
        rc, res, cig = ref.align(packed, ref_size, blob, off, lens, 1, tasks, 0, 1, scoring=sc, cigar_stride=stride)
        np.savez_compressed(os.path.join(HERE, "align_refine_planted_m150_sc_%s.npz" % sc_name), kind="align", blob=blob,
                            off=off, lens=lens, encoding=1, last_symbol=0, refine=1, tasks=tasks, results=res, cigars=cig,
                            cigar_stride=stride, scoring=scv)
        print("align_refine_planted", sc_name, "refined", int(res["refined"].sum()))
        _, _, (blob, off, lens), tasks = common.indel_workload(600, 150, seed + 2, gen=(codes, packed))
        prm = bindings.default_indel_params(0.6)
        prm.scoring = sc
        rc, res = ref.indel_match(packed, blob, off, lens, 0, tasks, prm)
        ct = common.indel_cigar_tasks(tasks, res)
        rc, cres, ccig = ref.indel_ext_cigar(packed, blob, off, lens, 0, ct, scoring=sc, cigar_stride=512)
        direction = (np.arange(len(res)) % 3 != 0).astype(np.uint8)
        rc, wire, sizes = ref.serialize(res, direction)
        np.savez_compressed(os.path.join(HERE, "indel_m150_sc_%s.npz" % sc_name), kind="indel", blob=blob, off=off, lens=lens,
                            encoding=0, min_clipped_factor=0.6, tasks=tasks, results=res, cigar_tasks=ct,
                            cigar_results=cres, cigars=ccig, cigar_stride=512, direction=direction, wire=wire,
                            wire_sizes=sizes, scoring=scv)
        print("indel", sc_name, "found", int(res["found"].sum()), "insertions", int((ct["indel_len1"] < 0).sum()))
        rec, cig = common.clip_workload(600, 150, seed + 3, gen=(codes, packed), scoring=sc)
        rc, rec2, cig2 = ref.clip_boundary(rec, cig, sc)
        np.savez_compressed(os.path.join(HERE, "clip_m150_sc_%s.npz" % sc_name), kind="clip", records=rec, cigars=cig,
                            out_records=rec2, out_cigars=cig2, scoring=scv)
        print("clip", sc_name, "changed", int((cig2 != cig).any(axis=1).sum()), "of", len(rec))
        t, rec, cig = common.clip_illegal_workload(orc, 500, seed=seed + 4, scoring=sc)
        rc, rec2, cig2 = ref.clip_illegal(t, rec, cig, sc)
        assert rc == 0
        np.savez_compressed(os.path.join(HERE, "clipillegal_m150_sc_%s.npz" % sc_name), kind="clipillegal", tasks=t,
                            records=rec, cigars=cig, out_records=rec2, out_cigars=cig2, scoring=scv)
        print("clipillegal", sc_name, "clipped", int((rec2["no_clippings"] > 0).sum()), "of", len(rec))
        _, _, (blob, off, lens), tasks = common.softclip_workload(400, seed=seed + 5, W=700, encoding=1, gen=(codes, packed))
        prm = bindings.default_softclip_params(fresh_object=0)
        rc, res, cig = ref.softclip_match(packed, blob, off, lens, 1, tasks, prm, sc, cigar_stride=384)
        assert rc == 0
        np.savez_compressed(os.path.join(HERE, "softclip_m150_w700_sc_%s.npz" % sc_name), kind="softclip", blob=blob, off=off,
                            lens=lens, encoding=1, tasks=tasks, results=res, cigars=cig, cigar_stride=384, fresh_object=0,
                            max_query_len=151, max_mismatches=12, scoring=scv)
        print("softclip", sc_name, "found", int(res["found"].sum()))
        (_, _), (blob, off, lens), desc, ridx, maps, direct, distant, stale = common.finish_workload(
            orc, 250, seed + 6, m=150, gen=(codes, packed), scoring=sc)
        _, wire, sizes = ref.serialize(maps, direct)
        rc, tasks, res, ext, cig, mdz = ref.finish_pairs(packed, ref_size, desc, blob, off, lens, ridx, wire, distant,
                                                         stale, scoring=sc, ext_stride=512)
        assert rc == 0 and res["ok"].all()
        w = [int(res[f].max()) + 1 for f in ("ext_cigar_len", "cigar_len", "mdz_len")]
        np.savez_compressed(os.path.join(HERE, "finish_m150_sc_%s.npz" % sc_name), kind="finish", blob=blob, off=off,
                            lens=lens, desc=desc, read_idx=ridx, wire=wire, wire_sizes=sizes, distant=distant,
                            stale_penalty=stale, tasks=tasks, results=res, ext_cigars=ext[:, :w[0]], cigars=cig[:, :w[1]],
                            mdz=mdz[:, :w[2]], flank=6, ext_stride=512, scoring=scv)
        print("finish", sc_name, "methods", np.bincount(res["method"], minlength=27))


def make_sa_loops(ref):
    """Round 3: what the reference's own suffix-array loops (CMappingCore::search_in_unknown_range_dir_and_rc /
    search_in_known_range_dir_and_rc, mapping_core_sa.cpp:193-513, with binary_sa_search and check_matching_part) push
    to dir / rc_match_positions for sorted reads over a genome with near-repeats, and the bytes the reference's heap
    gatherer + results collector make of those pushes.  The fixture carries one task of every kind per suffix-array entry
    the loops verified (tests/common.py: sa_loop_tasks), so the GPU box needs neither the suffix array nor the reference."""
    orc = bindings.Oracle()
    for name, seed, m, k, segi, sensitive in (("saloop_m150_k5_seg2", 8301, 150, 5, 2, 0),
                                              ("saloop_m150_k12_seg3_sens", 8302, 150, 12, 3, 1),
                                              ("saloop_m100_k2_seg1", 8303, 100, 2, 1, 0)):
        stage = int(k / 2.5) if sensitive else k
        seg = common.stage_segments(m, stage)[segi]
        codes, packed, reads, sa_dir, sa_rc = common.sa_loop_workload(seed, 260, 3, m, k, seg, min_read_len=m)
        ref_size = codes.numel()
        prm = bindings.default_indel_params(0.6)
        rc, ranges, best, pushed, pdir, pread = ref.search_reads(packed, ref_size, *reads, sa_dir, sa_rc, seg, m, k,
                                                                 sensitive=sensitive, enable_indels=sensitive, params=prm)
        assert rc == 0
        n = len(reads[2])
        reads2 = common.with_reverse_reads(reads)
        cf, dist, lev, indel, commit, read_first = common.sa_loop_tasks(ranges, reads[2], sa_dir, sa_rc, ref_size, seg, m, k, 50)
        # the reference's pushes as a per-candidate array (found = pushed), for its own collector
        committed = np.zeros(len(commit), bindings.MAPPING)
        cand_read = np.repeat(np.arange(n), np.diff(read_first).astype(np.int64))
        j = 0
        for i in range(len(commit)):      # a push belongs to the next candidate of its read and strand with its position
            if j < len(pushed) and pread[j] == cand_read[i] and pdir[j] == commit["direct"][i] and _same_locus(pushed[j], commit[i], m):
                committed[i] = pushed[j]
                committed[i]["found"] = 1
                j += 1
            else:
                committed[i]["type"] = 8
        assert j == len(pushed), (j, len(pushed))
        read_ids = (np.arange(n, dtype=np.uint64) * 3 + (5 << 32)).astype(np.uint64)
        gp = bindings.GroupParams(5, 4, 1024, 0)
        commit_full = commit.copy()
        commit_full["has_indel"] = sensitive
        rc, group = ref.result_groups(commit_full, committed, read_first, read_ids, gp)
        assert rc == 0
        blob, off, lens = reads2
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="saloop", packed=packed, ref_size=np.uint32(ref_size),
                            blob=blob, off=off, lens=lens, n_reads=n, seg=np.array(seg, np.uint32), k=k, min_read_len=m,
                            sensitive=sensitive, min_clipped_factor=0.6, cf=cf, dist=dist, lev=lev, indel=indel,
                            commit=commit, read_first=read_first, read_ids=read_ids, ranges=ranges, best=best,
                            pushed=pushed, pushed_direct=pdir, pushed_read=pread, group_bytes=group)
        print(name, "reads", n, "candidates", len(commit), "pushed", len(pushed), "non-lev", int((pushed["type"] != 0).sum()),
              "group bytes", len(group))


def _same_locus(push, task, m):
    """Does this push of the loops belong to this candidate?  A lev mapping carries the candidate's own position; Match's
    mapping lies within max_approx_indel_len of it (and m - 1 further on the other strand, mapping_core_sa.cpp:320)."""
    if push["type"] == 0:
        return int(push["pos"]) == int(task["lev_pos"])
    return abs(int(push["pos"]) - int(task["lev_pos"])) <= m + 64


def main():
    ref = bindings.Reference()
    codes = synth.make_genome(GENOME_SYMBOLS, seed=GENOME_SEED)
    packed = synth.pack_nibbles(codes).numpy()
    ref_size = codes.numel()
    if len(sys.argv) > 1 and sys.argv[1] == "groups":
        make_groups(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "finish":
        make_finish(ref, codes, packed, ref_size)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "rescue":
        make_rescue(ref, codes, packed, ref_size)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "saloop":
        make_sa_loops(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "scoring":
        make_scoring(ref, codes, packed, ref_size)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "round2":
        make_round2(ref, codes, packed, ref_size)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "cf":   # only the fixtures of this family
        make_cf(ref, codes, packed, ref_size)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "samcigar":
        make_sam_cigar(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "softclip":
        make_softclip(ref, codes, packed)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "clipl":
        make_clip_l(ref)
        return
    np.savez_compressed(os.path.join(HERE, "genome.npz"), packed=packed,
                        ref_size=np.uint32(ref_size), reference_version=ref.version())
    seed = 1000
    for name, m, k, W, variant, enc, last, n in DISTANCE:
        seed += 1
        rd = synth.make_reads(codes, n, m, k, seed)
        blob, off, lens = common.host_reads(rd, enc)
        tasks = synth.distance_tasks(rd, m, k, W, variant)
        rc, res = ref.distance(packed, ref_size, blob, off, lens, enc, tasks, last)
        assert rc == 0
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="distance", blob=blob, off=off,
                            lens=lens, encoding=enc, last_symbol=last, tasks=tasks, results=res)
        print(name, "accepted", int((res["edit_distance"] <= k).sum()), "of", n)
    for name, m, k, j, n in LEVDIAG:
        seed += 1
        fl, fr = common.stage_segments(m, k)[j]
        rd = synth.make_reads(codes, n, m, k, seed, rc_frac=0.0, protect=(fl, fr))
        blob, off, lens = common.host_reads(rd, 0)
        tasks = synth.levdiag_tasks(rd, k, fl, fr, 0)
        rc, res = ref.levdiag(packed, ref_size, 0, blob, off, lens, tasks)
        assert rc == 0
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="levdiag", blob=blob, off=off,
                            lens=lens, tasks=tasks, results=res, cur_offset=0)
        print(name, "accepted", int((res <= k).sum()), "of", n)
    for name, m, k, W, last, refine, n in ALIGN:
        seed += 1
        rd = synth.make_reads(codes, n, m, k, seed)
        blob, off, lens = common.host_reads(rd, 1)
        tasks = synth.distance_tasks(rd, m, k, W, 0)
        stride = 4 * ((2 * (m + k) + 8 + 3) // 4)
        rc, res, cig = ref.align(packed, ref_size, blob, off, lens, 1, tasks, last, refine,
                                 cigar_stride=stride)
        assert rc == 0
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="align", blob=blob, off=off,
                            lens=lens, encoding=1, last_symbol=last, refine=refine, tasks=tasks,
                            results=res, cigars=cig, cigar_stride=stride)
        print(name, "ok", int(res["ok"].sum()), "refined", int(res["refined"].sum()), "of", n)
    for name, m, k, n in [("align_refine_planted_m150", 150, 12, 400), ("align_refine_planted_m100", 100, 8, 200)]:
        seed += 1
        _, _, (blob, off, lens), tasks = common.planted_refine_workload(n, m, k, seed, gen=(codes, packed))
        stride = 4 * ((2 * (m + k) + 8 + 3) // 4)
        rc, res, cig = ref.align(packed, ref_size, blob, off, lens, 1, tasks, 0, 1, cigar_stride=stride)
        assert rc == 0
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="align", blob=blob, off=off,
                            lens=lens, encoding=1, last_symbol=0, refine=1, tasks=tasks,
                            results=res, cigars=cig, cigar_stride=stride)
        print(name, "ok", int(res["ok"].sum()), "refined", int(res["refined"].sum()), "of", n)
    for name, enc, mcf, n in [("indel_m150_nibble", 0, 1.0, 800), ("indel_m150_ascii_clip06", 1, 0.6, 800)]:
        seed += 1
        _, _, (blob, off, lens), tasks = common.indel_workload(n, 150, seed, encoding=enc, gen=(codes, packed))
        prm = bindings.default_indel_params(mcf)
        rc, res = ref.indel_match(packed, blob, off, lens, enc, tasks, prm)
        ct = common.indel_cigar_tasks(tasks, res)
        rc, cres, ccig = ref.indel_ext_cigar(packed, blob, off, lens, enc, ct, cigar_stride=512)
        direction = (np.arange(len(res)) % 3 != 0).astype(np.uint8)
        rc, wire, sizes = ref.serialize(res, direction)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="indel", blob=blob, off=off, lens=lens,
                            encoding=enc, min_clipped_factor=mcf, tasks=tasks, results=res,
                            cigar_tasks=ct, cigar_results=cres, cigars=ccig, cigar_stride=512,
                            direction=direction, wire=wire, wire_sizes=sizes)
        print(name, "found", int(res["found"].sum()), "types", np.bincount(res["type"][res["found"] == 1], minlength=9))
    seed += 1
    rec, cig = common.clip_workload(1200, 150, seed, gen=(codes, packed))
    make_cf(ref, codes, packed, ref_size)
    make_sam_cigar(ref)
    make_clip_l(ref)
    make_softclip(ref, codes, packed)
    rc, rec2, cig2 = ref.clip_boundary(rec, cig)
    np.savez_compressed(os.path.join(HERE, "clip_m150.npz"), kind="clip", records=rec, cigars=cig,
                        out_records=rec2, out_cigars=cig2)
    print("clip_m150 changed", int((cig2 != cig).any(axis=1).sum()), "of", len(rec))
    make_round2(ref, codes, packed, ref_size)
    make_scoring(ref, codes, packed, ref_size)
    make_sa_loops(ref)


if __name__ == "__main__":
    main()
