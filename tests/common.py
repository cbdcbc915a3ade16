"""Shared workload builders for the parity tests (seeded, small)."""
import numpy as np

from whisper_amd import synth

_GENOMES = {}

# non-default scoring_t values (user flags -match, -mismatch, -gap-*, -clipping: mapper/params.h:111-192):
# "asym": insertion and deletion costs differ (SURVEY appendix A hazards 8, 9); "odd": no value is a multiple
# of 0.2, so sums are not exact in binary and one decimal of AS:i: can round either way (hazard 11)
SCORING_VALUES = {
    "asym": (2.0, -3.0, -4.0, -0.6, -6.0, -0.2, -4.0),
    "odd": (1.0, -4.3, -5.7, -0.37, -4.9, -0.53, -7.1),
}


def scoring(name, module=None):
    """A Scoring struct (oracle.bindings' by default, whisper_amd.api's when module=api) or None for the default."""
    if name in (None, "default"):
        return None
    if module is None:
        from oracle import bindings as module
    return module.Scoring(*SCORING_VALUES[name])


def genome(n_symbols=1 << 20, seed=1):
    key = (n_symbols, seed)
    if key not in _GENOMES:
        codes = synth.make_genome(n_symbols, seed=seed)
        _GENOMES[key] = (codes, synth.pack_nibbles(codes).numpy())
    return _GENOMES[key]


def host_reads(rd, encoding):
    blob, off, lens = synth.pack_reads(rd["codes"], encoding)
    return blob.numpy(), off.numpy().astype(np.uint64), lens.numpy().astype(np.uint32)


def distance_workload(m, k, n, seed, encoding=0, W=None, variant=0, **kw):
    codes, packed = genome()
    rd = synth.make_reads(codes, n, m, k, seed, **kw)
    blob, off, lens = host_reads(rd, encoding)
    tasks = synth.distance_tasks(rd, m, k, W, variant)
    return codes, packed, (blob, off, lens), tasks


def low_complexity_genome(n_symbols=1 << 18, seed=5):
    """Short tandem repeats (period 1..6) and homopolymers between random stretches: the
    sequences on which alignments are ambiguous and band / tie rules show."""
    import torch
    key = ("lc", n_symbols, seed)
    if key not in _GENOMES:
        g = torch.Generator().manual_seed(seed)
        codes = synth.make_genome(n_symbols, seed=seed)
        lo, hi = synth.BOUNDARY_N_LEN, codes.numel() - synth.BOUNDARY_N_LEN
        pos = lo
        while pos < hi:
            ln = int(torch.randint(10, 160, (1,), generator=g))
            ln = min(ln, hi - pos)
            if int(torch.randint(0, 3, (1,), generator=g)):      # two thirds repetitive
                per = int(torch.randint(1, 7, (1,), generator=g))
                unit = torch.randint(0, 4, (per,), generator=g, dtype=torch.uint8)
                codes[pos:pos + ln] = unit.repeat(ln // per + 1)[:ln]
            pos += ln
        _GENOMES[key] = (codes, synth.pack_nibbles(codes).numpy())
    return _GENOMES[key]


def levdiag_workload(m, k, fl, fr, n, seed, cur_offset=0, gen=None):
    codes, packed = gen if gen is not None else genome()
    rd = synth.make_reads(codes, n, m, k, seed, rc_frac=0.0, protect=(fl, fr))
    blob, off, lens = host_reads(rd, 0)
    tasks = synth.levdiag_tasks(rd, k, fl, fr, cur_offset)
    return codes, packed, (blob, off, lens), tasks


def levdiag_odd_workload(n, seed):
    """LevDiag tasks the stage segments never produce (see test_levdiag_odd_shapes)."""
    from whisper_amd import api
    import torch
    codes, packed = genome()
    g = torch.Generator().manual_seed(seed)
    n_pos = torch.nonzero(codes[20000:900000] > 3).reshape(-1) + 20000
    assert n_pos.numel() > 100
    reads_codes, tasks = [], []
    for i in range(n):
        m = int(torch.randint(24, 161, (1,), generator=g))
        k = i % 7
        fl = int(torch.randint(0, m + 1, (1,), generator=g)) if i % 5 else 0
        fr = min(m, fl + int(torch.randint(0, 31, (1,), generator=g)))
        if i % 11 == 0:
            fr = m
        if i % 4 == 0:    # on an N run
            p = int(n_pos[int(torch.randint(0, n_pos.numel(), (1,), generator=g))]) - int(torch.randint(0, m, (1,), generator=g))
        else:
            p = int(torch.randint(20000, 900000, (1,), generator=g))
        r = codes[p:p + m].clone()
        for _ in range(int(torch.randint(0, k + 2, (1,), generator=g))):
            z = int(torch.randint(0, m, (1,), generator=g))
            if not (fl <= z < fr) and int(r[z]) < 4:
                r[z] = (int(r[z]) + 1) & 3
        if i % 8 < 4:
            r[r > 3] = 4
        reads_codes.append(r)
        tasks.append((i, p, fl, fr, k, 1))
    blob, off, lens = mixed_reads(reads_codes, 0)
    tasks = np.array(tasks, api.LEVDIAG_TASK)
    return codes, packed, (blob, off, lens), tasks


stage_segments = synth.stage_segments


def mixed_reads(codes_list, encoding):
    """Variable-length reads -> (blob, offsets, lens)."""
    import torch
    blobs, offs, lens = [], [0], []
    for c in codes_list:
        c = torch.as_tensor(c, dtype=torch.uint8)[None, :]
        b = (synth.pack_nibbles(c) if encoding == 0 else synth.to_ascii(c)).reshape(-1).numpy()
        blobs.append(b)
        offs.append(offs[-1] + len(b))
        lens.append(c.shape[1])
    return (np.concatenate(blobs), np.array(offs, np.uint64), np.array(lens, np.uint32))


def planted_refine_workload(n, m, k, seed, gen=None):
    """Reads carrying 1D2M1I / 1I2M1D next to other edits: what refine_ext_cigar
    (sam_matching.cpp:858-924) rewrites into three mismatches."""
    import torch
    codes, packed = gen if gen is not None else genome()
    g = torch.Generator().manual_seed(seed)
    G = codes.numel()
    out = torch.zeros((n, m), dtype=torch.uint8)
    ps = torch.randint(synth.BOUNDARY_N_LEN, G - synth.BOUNDARY_N_LEN - m - 16, (n,), generator=g)
    is_rc = torch.rand(n, generator=g) < 0.5
    for i in range(n):
        p = int(ps[i])
        src = codes[p:p + m + 8].clone()
        src[src > 3] = 0
        q = int(torch.randint(8, m - 16, (1,), generator=g))
        x = (int(src[q + 3]) + 1 + int(torch.randint(0, 3, (1,), generator=g))) & 3
        if i & 1:   # deletion of genome symbol q, insertion two matches later
            r = torch.cat([src[:q], src[q + 1:q + 3], torch.tensor([x], dtype=torch.uint8), src[q + 3:]])
        else:       # insertion first, deletion two matches later
            r = torch.cat([src[:q], torch.tensor([x], dtype=torch.uint8), src[q:q + 2], src[q + 3:]])
        r = r[:m].clone()
        if i % 3 == 0:   # one more substitution somewhere else
            z = (q + 30) % m
            r[z] = (int(r[z]) + 1) & 3
        out[i] = synth.rev_comp(r) if bool(is_rc[i]) else r
    rd = dict(codes=out, p=ps, is_true=torch.ones(n, dtype=torch.bool), is_rc=is_rc)
    blob, off, lens = host_reads(rd, 1)
    tasks = synth.distance_tasks(rd, m, k, None, 0)
    return codes, packed, (blob, off, lens), tasks


def indel_workload(n, m, seed, max_indel=50, max_mm=12, stage_major=5, encoding=0, gen=None):
    """Candidates for CIndelMatching::Match: reads carrying one long indel (3..max_indel) on
    either side of the anchor segment, reads with a junk end (clipping), plain reads with a
    few substitutions, and reads where only the anchor matches."""
    import torch
    from oracle import bindings as ob
    codes, packed = gen if gen is not None else genome()
    g = torch.Generator().manual_seed(seed)
    G = codes.numel()
    segs = stage_segments(m, stage_major)
    out = torch.zeros((n, m), dtype=torch.uint8)
    tasks = np.zeros(n, ob.INDEL_TASK)

    def rnd(lo, hi):
        return int(torch.randint(lo, hi, (1,), generator=g))

    clean = codes.clone()
    for i in range(n):
        fl, fr = segs[rnd(0, len(segs))]
        p = rnd(synth.BOUNDARY_N_LEN + 200, G - synth.BOUNDARY_N_LEN - m - 200)
        base = clean[p:p + m].clone()
        kind = i % 8
        L = rnd(3, max_indel + 1)
        r = base.clone()
        if kind in (0, 1) and fl > 12:          # indel left of the anchor
            x = rnd(8, fl)                      # read[x:] stays aligned at p
            if kind == 0:                       # deletion: the prefix comes from L earlier
                r[:x] = clean[p - L:p - L + x]
            else:                               # insertion of L (or fewer) random symbols
                Li = min(L, x - 7) if x - 7 >= 3 else 0
                if Li:
                    r[:x - Li] = clean[p + Li:p + x]
                    r[x - Li:x] = torch.randint(0, 4, (Li,), generator=g, dtype=torch.uint8)
        elif kind in (2, 3) and m - fr > 12:    # indel right of the anchor
            x = rnd(fr, m - 8)
            if kind == 2:
                r[x:] = clean[p + x + L:p + m + L]
            else:
                Li = min(L, m - x - 7) if m - x - 7 >= 3 else 0
                if Li:
                    r[x:x + Li] = torch.randint(0, 4, (Li,), generator=g, dtype=torch.uint8)
                    r[x + Li:] = clean[p + x:p + m - Li]
        elif kind == 4 and fl > 25:             # junk at the left end
            c = rnd(10, fl - 5)
            r[:c] = torch.randint(0, 4, (c,), generator=g, dtype=torch.uint8)
        elif kind == 5 and m - fr > 25:         # junk at the right end
            c = rnd(10, m - fr - 5)
            r[m - c:] = torch.randint(0, 4, (c,), generator=g, dtype=torch.uint8)
        elif kind == 6:                         # both: indel on the left, junk on the right
            if fl > 12:
                x = rnd(8, fl)
                r[:x] = clean[p - L:p - L + x]
            if m - fr > 25:
                c = rnd(10, m - fr - 5)
                r[m - c:] = torch.randint(0, 4, (c,), generator=g, dtype=torch.uint8)
        else:                                   # only the anchor is shared
            if i % 16 == 7:
                r = torch.randint(0, 4, (m,), generator=g, dtype=torch.uint8)
                r[fl:fr] = base[fl:fr]
        for _ in range(rnd(0, 4)):              # a few substitutions outside the anchor
            z = rnd(0, m)
            if not (fl <= z < fr):
                r[z] = (int(r[z]) + rnd(1, 4)) & 3
        r[r > 3] = 4
        if i % 31 == 0:
            r[fl:fr] = base[fl:fr]
            r[(fl + 3) % m if not (fl <= (fl + 3) % m < fr) else 0] = 4
        out[i] = r
        tasks[i] = (i, p, max_indel, fl, fr - fl, max_mm, 0)
    rd = dict(codes=out)
    blob, off, lens = host_reads(rd, encoding)
    return codes, packed, (blob, off, lens), tasks


def indel_cigar_tasks(match_tasks, mappings):
    """fill_cigar_with_indel_matching inputs for the mappings Match found (direct strand:
    examined_start_pos = mapping.pos, sam_matching.cpp:826-828)."""
    from oracle import bindings as ob
    idx = np.nonzero(mappings["found"])[0]
    t = np.zeros(len(idx), ob.INDEL_CIGAR_TASK)
    t["read_idx"] = match_tasks["read_idx"][idx]
    t["examined_start_pos"] = mappings["pos"][idx]
    for f in ("type", "matching_len1", "matching_len2", "indel_len1", "indel_len2", "clipping_len"):
        t[f] = mappings[f][idx]
    return t


def clip_workload(n, m, seed, gen=None, scoring=None):
    """ext-CIGARs with noisy ends (what clipBoundaryPositions clips) produced by aligning
    reads whose first / last symbols are junk; returns (records, cigars) as the caller of
    clipBoundaryPositions holds them after fill_cigar_with_lev."""
    import torch
    from oracle import bindings as ob
    codes, packed = gen if gen is not None else genome()
    g = torch.Generator().manual_seed(seed)
    k = 40
    rd = synth.make_reads(codes, n, m, 4, seed, true_frac=1.0, rc_frac=0.3)
    r = rd["codes"]
    for i in range(n):
        side = i % 4
        c = int(torch.randint(8, 30, (1,), generator=g))
        junk = torch.randint(0, 4, (c,), generator=g, dtype=torch.uint8)
        if side in (0, 2):
            r[i, :c] = junk
        if side in (1, 2):
            c2 = int(torch.randint(8, 30, (1,), generator=g))
            r[i, m - c2:] = torch.randint(0, 4, (c2,), generator=g, dtype=torch.uint8)
    blob, off, lens = host_reads(rd, 1)
    tasks = synth.distance_tasks(rd, m, k)
    orc = ob.Oracle()
    stride = 4 * ((2 * (m + k) + 8 + 3) // 4)
    rc, res, cig = orc.align(packed, blob, off, lens, 1, tasks, 0, 1, cigar_stride=stride, scoring=scoring)
    ok = np.nonzero(res["ok"])[0]
    rec = np.zeros(len(ok), ob.CLIP_RECORD)
    rec["score"] = res["affine_score"][ok]
    rec["ref_seq_pos"] = 1000 + res["begin_pos"][ok]
    rec["ref_length"] = (res["end_pos"][ok] - res["begin_pos"][ok]).astype(np.uint16)
    rec["err_edit_distance"] = res["edit_distance"][ok]
    return rec, np.ascontiguousarray(cig[ok])


def cf_workload(n, seed, min_len=150, k=5, cur_offset=0, gen=None):
    """CF pre-filter tasks (mapping_core_sa.cpp:228-230 / 283-285): reads of min_len .. min_len + 5
    symbols, direct-strand and other-strand flavours, every segment of stage k as the anchor."""
    import torch
    from oracle import bindings as ob
    codes, packed = gen if gen is not None else genome()
    g = torch.Generator().manual_seed(seed)
    ref_size = codes.numel()
    segs = stage_segments(min_len, k)
    reads_codes, tasks = [], np.zeros(n, ob.CF_TASK)
    for i in range(n):
        m = min_len + int(torch.randint(0, 6, (1,), generator=g))
        p = int(torch.randint(synth.BOUNDARY_N_LEN - 40, ref_size - synth.BOUNDARY_N_LEN - 400, (1,), generator=g))
        r = codes[p:p + m].clone()
        kind = i % 4
        if kind == 1:      # a few substitutions
            for _ in range(int(torch.randint(1, 8, (1,), generator=g))):
                z = int(torch.randint(0, m, (1,), generator=g))
                r[z] = (int(r[z]) + 1 + int(torch.randint(0, 3, (1,), generator=g))) & 3
        elif kind == 2:    # unrelated read
            r = torch.randint(0, 4, (m,), generator=g, dtype=torch.uint8)
        elif kind == 3:    # low complexity
            r = torch.tensor([0, 1] * m, dtype=torch.uint8)[:m] if i % 8 == 3 else torch.zeros(m, dtype=torch.uint8)
        r[r > 3] = 4
        fl, fr = segs[int(torch.randint(0, len(segs), (1,), generator=g))]
        other = bool(i & 16)
        t = tasks[i]
        t["read_idx"] = i
        t["pattern_len"] = min_len
        if not other:      # direct strand: window = first min_len symbols, left_end = position - cur_offset
            t["position"] = p + cur_offset
            t["read_offset"] = 0
            t["fixed_left"], t["fixed_right"] = fl, fr
            t["dir_gen_flag"] = 1
            t["flags"] = 0
            reads_codes.append(r)
        else:              # other strand: the mapper holds the reverse complement, window = its last min_len symbols
            t["position"] = (ref_size - p - min_len + cur_offset) & 0xFFFFFFFF
            t["read_offset"] = m - min_len
            t["fixed_left"], t["fixed_right"] = min_len - fr, min_len - fl
            t["dir_gen_flag"] = 0
            t["flags"] = 1   # stored direct, verified in rev_comp orientation
            # oriented read = rev_comp(stored); make its window equal the genome at p
            ext = codes[p - (m - min_len):p + min_len].clone() if kind != 2 else r
            ext[ext > 3] = 4
            if kind in (1, 3):
                ext = r
            reads_codes.append(synth.rev_comp(ext))
    blob, off, lens = mixed_reads(reads_codes, 0)
    return codes, packed, (blob, off, lens), tasks


def ext_cigar_pool(oracle, n=3000, seed=11, stride=512):
    """ext-CIGARs of every flavour the post-processing produces: Myers alignments with
    indels and refinement, the same after boundary clipping ('$'), and a few hand-made
    corner cases (empty, all clipped, adjacent deletions, insertion at either end)."""
    codes, packed, (blob, off, lens), tasks = distance_workload(150, 13, n, seed=seed, encoding=1, W=177,
                                                               true_frac=0.9)
    rc, res, cig = oracle.align(packed, blob, off, lens, 1, tasks, last_symbol=0, refine=1, cigar_stride=stride)
    pool = [cig[res["ok"] != 0]]
    rec, ccig = clip_workload(n // 3, 150, seed + 1)
    _, rec2, ccig2 = oracle.clip_boundary(rec, ccig)
    pad = np.zeros((ccig2.shape[0], stride), np.uint8)
    pad[:, :min(stride, ccig2.shape[1])] = ccig2[:, :stride]
    pool.append(pad)
    hand = [b"", b".", b"$$$$", b"^A^C..", b"#A....#C", b"..A^G^T#C#C..", b"$$..T..$", b"AC", b"#A", b"^A",
            b"....^A....", b".#T.", b"$.$"]
    hm = np.zeros((len(hand), stride), np.uint8)
    for i, h in enumerate(hand):
        hm[i, :len(h)] = np.frombuffer(h, np.uint8)
    pool.append(hm)
    return np.ascontiguousarray(np.concatenate(pool))


def clip_illegal_workload(oracle, n, seed, m=150, scoring=None):
    """Inputs of clipIllegalPositions (soft_clipping.cpp:592-729): aligned records (some already
    boundary-clipped, some perfect matches without an ext-CIGAR) placed so that they hang over the
    start of their chromosome, over its end, over both, or not at all."""
    from oracle import bindings as ob
    rec, cig = clip_workload(n, m, seed, scoring=scoring)
    _, rec_b, cig_b = oracle.clip_boundary(rec, cig, scoring)
    rng = np.random.default_rng(seed)
    n = len(rec)
    use_b = rng.random(n) < 0.3
    rec = np.where(use_b, rec_b, rec)
    cig = np.where(use_b[:, None], cig_b, cig)
    rec["no_clippings"] = 0
    tasks = np.zeros(n, ob.CLIP_ILLEGAL_TASK)
    tasks["read_length"] = m
    perfect = rng.random(n) < 0.15
    tasks["perfect"] = perfect
    cig[perfect] = 0
    rec["ref_length"][perfect] = m
    rec["err_edit_distance"][perfect] = 0
    rec["left_clipping"][perfect] = 0
    rec["right_clipping"][perfect] = 0
    rec["score"][perfect] = float(m) * (scoring.match if scoring is not None else 1.0)
    rec["cigar_len"] = [len(bytes(c).split(b"\0")[0]) for c in cig]
    kind = rng.integers(0, 4, n)
    over_l = rng.integers(0, 40, n)
    over_r = rng.integers(1, 40, n)
    pos = rec["ref_seq_pos"].astype(np.int64)
    pos = np.where((kind == 0) | (kind == 2), -over_l, pos)
    rec["ref_seq_pos"] = pos
    last = pos + rec["ref_length"].astype(np.int64) - 1
    tasks["ref_size"] = np.where((kind == 1) | (kind == 2), last - over_r, last + rng.integers(0, 1000, n))
    return tasks, rec, np.ascontiguousarray(cig)


def clip_pairs_workload(oracle, n_pairs, seed, m=150, scoring=None):
    """Inputs of clipOverlappingPairs (soft_clipping.cpp:997-1044): pairs of aligned records, most of
    them with the reverse read left of and overlapping the forward read, plus the cases the routine
    leaves alone (other chromosome, a clipped mate, proper order, no overlap) and perfect matches."""
    from oracle import bindings as ob
    rec, cig = clip_workload(2 * n_pairs + 64, m, seed, scoring=scoring)
    n_pairs = min(n_pairs, len(rec) // 2)
    rec, cig = rec[:2 * n_pairs].copy(), np.ascontiguousarray(cig[:2 * n_pairs])
    rng = np.random.default_rng(seed + 1)
    rec["no_clippings"] = 0
    tasks = np.zeros(n_pairs, ob.CLIP_PAIR_TASK)
    tasks["read_length"] = m
    for i in range(n_pairs):
        a, b = 2 * i, 2 * i + 1
        kind = i % 8
        for q in (a, b):
            if rng.random() < 0.15:
                tasks["perfect"][i][q - a] = 1
                cig[q] = 0
                rec["ref_length"][q] = m
                rec["err_edit_distance"][q] = 0
                rec["score"][q] = float(m) * (scoring.match if scoring is not None else 1.0)
        la = int(rec["ref_length"][a])
        swap = bool(rng.integers(0, 2))            # which of first / second is the left one
        l, r = (b, a) if swap else (a, b)
        ll = int(rec["ref_length"][l])
        base = int(rng.integers(100, 100000))
        rec["ref_seq_pos"][l] = base
        rec["ref_seq_pos"][r] = base + int(rng.integers(0, ll))     # overlap: left end > right start
        tasks["direct"][i][l - a] = 0
        tasks["direct"][i][r - a] = 1
        tasks["ref_seq_id"][i] = (3, 3)
        if kind == 1:
            tasks["ref_seq_id"][i] = (3, 4)
        elif kind == 2:
            rec["left_clipping" if i & 8 else "right_clipping"][a if i & 16 else b] = 5
        elif kind == 3:                                              # proper orientation: forward read on the left
            tasks["direct"][i][l - a] = 1
            tasks["direct"][i][r - a] = 0
        elif kind == 4:                                              # same start: `direct` decides who is left
            rec["ref_seq_pos"][r] = base
        elif kind == 5:                                              # no overlap
            rec["ref_seq_pos"][r] = base + ll + int(rng.integers(0, 50))
        del la
    rec["cigar_len"] = [len(bytes(c).split(b"\0")[0]) for c in cig]
    return tasks, rec, cig


def softclip_workload(n, seed, m=150, W=700, encoding=1, gen=None, lens_jitter=False):
    """Inputs of CSoftClipping::match (soft_clipping.cpp:234-495) as the mate rescue builds them
    (sam_matching.cpp:543-558): reads matching exactly except for junk ends (-> clipping), reads
    with one long deletion / insertion with and without a few substitutions around it, reads that
    share nothing with the window, in both orientations, anywhere inside a window of W symbols."""
    import torch
    from oracle import bindings as ob
    codes, packed = gen if gen is not None else genome()
    g = torch.Generator().manual_seed(seed)
    G = codes.numel()
    reads, tasks = [], np.zeros(n, ob.SOFTCLIP_TASK)

    def rnd(lo, hi):
        return int(torch.randint(lo, hi, (1,), generator=g))

    def junk(k):
        return torch.randint(0, 4, (k,), generator=g, dtype=torch.uint8)

    for i in range(n):
        mi = m - rnd(0, 8) if lens_jitter else m
        p = rnd(synth.BOUNDARY_N_LEN + W + 200, G - synth.BOUNDARY_N_LEN - 2 * W - 400)
        src = codes[p:p + mi + 300].clone()
        kind = i % 10
        if kind in (0, 1, 2):                      # exact core, junk at either end or both
            r = src[:mi].clone()
            if kind in (0, 2):
                c = rnd(1, 70)
                r[:c] = junk(c)
            if kind in (1, 2):
                c = rnd(1, 70)
                r[mi - c:] = junk(c)
        elif kind in (3, 4):                       # long deletion, (4) with substitutions around it
            x, L = rnd(22, mi - 22), rnd(1, 120)
            r = torch.cat([src[:x], src[x + L:x + L + mi - x]])
        elif kind in (5, 6):                       # long insertion
            x, L = rnd(22, mi - 45), rnd(1, 60)
            L = min(L, mi - x - 22)
            r = torch.cat([src[:x], junk(L), src[x:x + mi - x - L]])
        elif kind == 7:                            # nothing in common
            r = junk(mi)
        elif kind == 8:                            # two exact blocks from different places (long gap)
            x = rnd(25, mi - 25)
            r = torch.cat([src[:x], src[200:200 + mi - x]])
        else:                                      # untouched copy
            r = src[:mi].clone()
        if kind in (4, 6) or i % 7 == 0:
            for _ in range(rnd(1, 6)):
                z = rnd(0, mi)
                r[z] = (int(r[z]) + rnd(1, 4)) & 3
        r = r[:mi].clone()
        r[r > 3] = 4
        if i % 23 == 0:
            r[rnd(0, mi)] = 4
        rc = bool(i & 1)
        reads.append(synth.rev_comp(r) if rc else r)
        start = p - rnd(0, W - mi - 60)
        if i % 41 == 0:
            start = p                                 # match at the very first window position
        tasks[i] = (i, start, W, 20, 1 if rc else 0, 0)
    blob, off, lens = mixed_reads(reads, encoding)
    return codes, packed, (blob, off, lens), tasks


# ---- the chosen pair: records -> SAM fields (sam.cpp:1555-1580, 1871-1900, 1992-2007) ------------------
SEPARATE_N_LEN = 1536      # common/defs.h:43


def chromosomes(ref_size):
    """A chromosome table over the test genome: rows of (size, pos_in_ref_seq, no_initial_Ns, no_final_Ns,
    no_starting_Ns) as CRefSeqDesc::Load reads them (ref_desc.cpp:33-75).  The third one starts with trimmed Ns,
    so positions in its first symbols translate to ref_seq_pos <= 0."""
    b = synth.BOUNDARY_N_LEN
    usable = ref_size - 2 * b
    s1, s2 = usable // 3, usable // 3
    p1 = b
    p2 = p1 + s1 + SEPARATE_N_LEN
    p3 = p2 + s2 + SEPARATE_N_LEN
    s3 = ref_size - b - p3
    return np.array([[s1, p1, 7, 0, 0], [s2, p2, 0, 11, 0], [s3, p3, 0, 0, 900]], np.uint64)


def translate(desc, raw_pos):
    """CRefSeqDesc::Translate (ref_desc.cpp:102-140): (ok, ref_seq_id, ref_seq_pos)."""
    raw1 = raw_pos + 1
    keys = desc[:, 1].astype(np.int64) + SEPARATE_N_LEN // 2
    j = int(np.searchsorted(keys, raw1, side="left"))      # first row with key >= raw1
    if j == len(desc) and raw_pos >= int(desc[-1, 1]) + int(desc[-1, 0]) + 2 * SEPARATE_N_LEN:
        return False, 0, 0
    if j == 0:
        return False, 0, 0
    c = desc[j - 1].astype(np.int64)
    return True, j - 1, int(raw_pos - c[1] + c[2] - c[4] + 1)


def finish_tasks(desc, read_idx, lens, mappings, direct, distant, stale_penalty=None, scoring=None, isize=(-375, 1375)):
    """isize: ceil / floor of mean -+ getMyersDev of the insert-size model (default: the untrained model, 500 -+ 7 * 125).
    mapping_desc_t as store_mapped_pair_reads constructs it from a record (sam.cpp:1555-1580,
    common/types.h:371-425): the input of wamd_finish_pairs."""
    from whisper_amd import api
    if scoring is None:
        match, mismatch, ins_o, ins_e, del_o, del_e = 1.0, -5.0, -5.0, -0.4, -5.0, -0.4
    else:
        match, mismatch, ins_o, ins_e, del_o, del_e = (scoring.match, scoring.mismatch, scoring.gap_ins_open,
                                                       scoring.gap_ins_extend, scoring.gap_del_open, scoring.gap_del_extend)
    n = len(read_idx)
    t = np.zeros(n, api.FINISH_TASK)

    def i2m(size):     # indel_to_mismatch_score, types.h:342-352
        a = del_o + (size - 1) * del_e if size > 0 else ins_o + (-size - 1) * ins_e
        return int(a / mismatch) & 0xFFFF

    for i in range(n):
        mp = mappings[i]
        m = int(lens[read_idx[i]])
        ok, sid, spos = translate(desc, int(mp["pos"]))
        if ok and not direct[i]:
            spos -= m - 1
        if not ok:
            spos = 0
        ty, nm = int(mp["type"]), int(mp["no_mismatches"]) & 0x3f
        flags = 0
        if ty == 0:
            ed, pen = nm, int(stale_penalty[i]) if stale_penalty is not None else 0
        else:
            pen = nm + (1 if ty in (1, 5, 6) else 2)              # calc_penalty, results.cpp:327-343
            l1 = int(mp["indel_len1"]) if ty in (1, 2, 3, 4) else 0
            l2 = int(mp["indel_len2"]) if ty == 2 else 0
            ed = {1: nm + i2m(l1), 2: nm + i2m(l1) + i2m(l2), 7: nm + 2, 3: nm + 1 + i2m(l1), 4: nm + 1 + i2m(l1),
                  5: nm + 1, 6: nm + 1}[ty] & 0xFFFF
        if ty != 0 and not (ed == 0 and pen == 0):
            # the non-lev types end up where Translate puts examined_start_pos (sam_matching.cpp:826-828, 843-849)
            ok2, sid2, spos2 = translate(desc, int(mp["pos"]) if direct[i] else int(mp["pos"]) - (m - 1))
            if ok2:
                sid, spos = sid2, spos2
            else:
                flags = 1      # WAMD_FINISH_NO_TRANSLATE: mapping.pos and the chromosome stay (sam_matching.cpp:843-849)
        # what the wire record carries per type (results.cpp:161-275): the rest deserialises as 0
        carried = {0: (), 1: ("matching_len1", "indel_len1"), 2: ("matching_len1", "matching_len2", "indel_len1", "indel_len2"),
                   3: ("matching_len1", "clipping_len", "indel_len1"), 4: ("matching_len1", "clipping_len", "indel_len1"),
                   5: ("matching_len1",), 6: ("matching_len1",), 7: ("matching_len1", "clipping_len")}[ty]
        v = {f: (int(mp[f]) if f in carried else 0)
             for f in ("matching_len1", "matching_len2", "indel_len1", "indel_len2", "clipping_len")}
        t[i] = (read_idx[i], mp["pos"], ty, pen, v["matching_len1"], v["matching_len2"], v["indel_len1"],
                v["indel_len2"], v["clipping_len"], sid, spos, int(desc[sid, 0] + desc[sid, 2] + desc[sid, 3]),
                ed, direct[i], distant[i], isize[0], isize[1], flags)
    return t


FINISH_MODEL = (330.0, 12.0)       # an insert-size model that calls part of finish_workload's pairs distant
FINISH_ISIZE = (246, 414)          # ceil(mean - 7 dev), floor(mean + 7 dev): getMyersDev = 7 sigma (sam.cpp:185-188)


def finish_workload(oracle, n_pairs, seed, m=150, flank=6, gen=None, scoring=None, from_pair_frac=0.0):
    """Pairs as store_mapped_pair_reads holds them once it has picked one: perfect matches, lev records with
    substitutions / short indels / the refinement pattern, indel-matcher records of every type, junk ends (boundary
    clipping), mappings over a chromosome's end or before its first base (illegal positions), reverse reads left of
    and overlapping their forward mates (paired clipping).  Returns (genome, reads, desc, read_idx, mappings, direct,
    distant, stale_penalty)."""
    import torch
    from oracle import bindings as ob
    codes, packed = gen if gen is not None else genome()
    G = codes.numel()
    desc = chromosomes(G)
    g = torch.Generator().manual_seed(seed)

    def rnd(lo, hi):
        return int(torch.randint(lo, hi, (1,), generator=g))

    # a pool of indel-matcher records (direct-strand Match on reads carrying one long indel or junk ends)
    _, _, (iblob, ioff, ilens), itasks = indel_workload(max(64, n_pairs), m, seed + 1, gen=(codes, packed))
    iprm = ob.default_indel_params(0.6)
    if scoring is not None:
        iprm.scoring = scoring
    _, imap = oracle.indel_match(packed, iblob, ioff, ilens, 0, itasks, iprm)
    ipool = [j for j in np.nonzero(imap["found"])[0]]
    icodes = np.zeros((len(ilens), m), np.uint8)
    ib = iblob.reshape(len(ilens), -1)
    icodes[:, 0::2] = (ib >> 4)[:, :(m + 1) // 2]
    icodes[:, 1::2] = (ib & 15)[:, :m // 2]

    clean = codes.clone()
    clean[clean > 3] = 0
    reads, mappings, direct = [], np.zeros(2 * n_pairs, ob.MAPPING), np.zeros(2 * n_pairs, np.uint8)
    want_lev = []

    def mutate(r, kind):
        r = r.clone()
        if kind == "subs":
            for _ in range(rnd(1, 5)):
                z = rnd(0, m)
                r[z] = (int(r[z]) + rnd(1, 4)) & 3
        elif kind == "indel":
            q = rnd(10, m - 20)
            if rnd(0, 2):
                r = torch.cat([r[:q], r[q + 2:], torch.randint(0, 4, (2,), generator=g, dtype=torch.uint8)])
            else:
                r = torch.cat([r[:q], torch.randint(0, 4, (1,), generator=g, dtype=torch.uint8), r[q:m - 1]])
        elif kind == "refine":      # 1I2M1D: what refine_ext_cigar rewrites
            q = rnd(10, m - 20)
            x = (int(r[q + 3]) + 1) & 3
            r = torch.cat([r[:q], torch.tensor([x], dtype=torch.uint8), r[q:q + 2], r[q + 3:]])[:m]
        elif kind == "junk":
            c = rnd(8, 30)
            if rnd(0, 2):
                r[:c] = torch.randint(0, 4, (c,), generator=g, dtype=torch.uint8)
            else:
                r[m - c:] = torch.randint(0, 4, (c,), generator=g, dtype=torch.uint8)
            z = rnd(40, m - 40)
            r[z] = (int(r[z]) + 1) & 3
        return r[:m].clone()

    for i in range(n_pairs):
        kind = i % 10
        while True:
            c = rnd(0, 3)
            c_pos, c_size = int(desc[c, 1]), int(desc[c, 0])
            f = c_pos + rnd(1000, c_size - 2000)
            isize = rnd(m + 60, 520)
            pf, pr = f, f + isize - m                    # forward read's start, reverse read's start
            if kind in (2, 6):                           # reverse read left of and overlapping the forward read
                pr = f
                pf = f + rnd(0, m - 10)
            if kind == 5:                                # over the end of the chromosome / before its first base
                if c == 2 and rnd(0, 2):
                    pf = c_pos + rnd(780, 905)
                    pr = pf + isize - m
                else:
                    pr = c_pos + c_size - m + rnd(1, 40)
                    pf = pr - (isize - m)
            if int(codes[pf - 8:pf + m + 8].max()) <= 3 and int(codes[pr - 8:pr + m + 8].max()) <= 3:
                break                                    # no N run under either read
        first_forward = bool(rnd(0, 2))
        for r in range(2):
            slot = 2 * i + r
            fwd = (r == 0) == first_forward
            p = pf if fwd else pr
            src = clean[p:p + m + 4]
            if kind == 4 and r == 0 and ipool:           # an indel-matcher record
                j = ipool[(i // 10) % len(ipool)]
                q = torch.from_numpy(icodes[j].copy())
                mappings[slot] = imap[j]
                if fwd:
                    reads.append(q)
                else:
                    reads.append(synth.rev_comp(q))
                    mappings[slot]["pos"] = int(imap[j]["pos"]) + m - 1
                direct[slot] = fwd
                continue
            mk = {0: "subs", 1: "none", 2: "subs", 3: "junk", 4: "subs", 5: "subs", 6: "none", 7: "indel", 8: "refine",
                  9: "subs" if r else "none"}[kind]
            q = mutate(src[:m], mk) if mk != "none" else src[:m].clone()
            reads.append(q if fwd else synth.rev_comp(q))
            direct[slot] = fwd
            mappings[slot]["type"] = 0
            mappings[slot]["pos"] = p if fwd else p + m - 1
            want_lev.append(slot)
    blob, off, lens = mixed_reads(reads, 1)
    read_idx = np.arange(2 * n_pairs, dtype=np.uint32)
    # the error count the mapping stage recorded = the distance inside fill_cigar_with_lev's window
    lev = np.array(want_lev)
    dt = np.zeros(len(lev), ob.DIST_TASK)
    dt["read_idx"] = lev
    begin = np.where(direct[lev] == 1, mappings["pos"][lev], mappings["pos"][lev] - m + 1)
    dt["ref_pos"] = begin - flank
    dt["max_distance_in_ref"] = m + 2 * flank + 1
    dt["max_ed"] = 40
    dt["flags"] = 1 - direct[lev]
    _, dres = oracle.distance(packed, blob, off, lens, 1, dt, 0)
    assert (dres["edit_distance"] <= 40).all()
    mappings["no_mismatches"][lev] = dres["edit_distance"]
    mappings["found"] = 1
    rng = np.random.default_rng(seed)
    slack = rng.random(len(lev)) < 0.2                   # LevDiag's anchored distance can exceed the window's
    mappings["no_mismatches"][lev[slack]] += ((dres["edit_distance"][slack] > 0) *
                                              rng.integers(1, 3, slack.sum())).astype(np.uint32)
    distant = (rng.random(2 * n_pairs) < 0.1).astype(np.uint8)
    stale = np.zeros(2 * n_pairs, np.uint32)
    stale[lev] = np.where(rng.random(len(lev)) < 0.05, 3, 0)   # a lev record deserialised after a non-lev one
    if from_pair_frac > 0:     # isDistant left to the device (WAMD_DISTANT_FROM_PAIR), both slots of a pair
        rng2 = np.random.default_rng(seed + 77)
        distant[np.repeat(rng2.random(n_pairs) < from_pair_frac, 2)] = 2
    return (codes, packed), (blob, off, lens), desc, read_idx, mappings, direct, distant, stale


def same_finish_output(got, want, skip=()):
    """(results, ext-CIGARs, CIGARs, MD:Z) of two finish_pairs runs: every field and every string equal."""
    gr, ge, gc, gm = got
    wr, we, wc, wm = want
    for f in wr.dtype.names:
        if f not in ("reserved",) + tuple(skip):
            np.testing.assert_array_equal(gr[f], wr[f], err_msg=f)
    for i in range(len(wr)):
        for a, b, what in ((ge, we, "ext-CIGAR"), (gc, wc, "CIGAR"), (gm, wm, "MD:Z")):
            x, y = bytes(a[i]).split(b"\0")[0], bytes(b[i]).split(b"\0")[0]
            assert x == y, "%s of slot %d: %r != %r" % (what, i, x, y)


def commit_workload(oracle, n_reads, seed, k=12, max_cand=14):
    """Candidates as the SA loop holds them after verification (mapping_core_sa.cpp:241-344): per read a run of
    direct-strand candidates, then reverse-strand ones; cur_match around the threshold, positions with ties, a
    third of the candidates also carrying a CIndelMatching::Match result of any type."""
    from oracle import bindings as ob
    rng = np.random.default_rng(seed)
    _, packed, (blob, off, lens), itasks = indel_workload(600, 150, seed + 1)
    _, pool = oracle.indel_match(packed, blob, off, lens, 0, itasks, ob.default_indel_params(0.6))
    pool = pool[pool["found"] == 1]
    n_cand = rng.integers(0, max_cand + 1, n_reads)
    n_cand[rng.random(n_reads) < 0.02] = 60            # a few reads with more candidates than a small heap holds
    read_first = np.concatenate([[0], np.cumsum(n_cand)]).astype(np.uint32)
    n = int(read_first[-1])
    t = np.zeros(n, ob.COMMIT_TASK)
    indel = np.zeros(n, ob.MAPPING)
    t["max_mismatches"] = k
    t["pattern_len"] = 150
    t["lev_errors"] = rng.integers(0, k + 6, n)
    base = rng.integers(20000, 30000, n_reads)
    for r in range(n_reads):
        lo, hi = read_first[r], read_first[r + 1]
        c = hi - lo
        t["lev_pos"][lo:hi] = base[r] + rng.integers(0, 6, c) * 7          # few distinct positions: ties on (penalty, pos)
        n_dir = rng.integers(0, c + 1)
        t["direct"][lo:lo + n_dir] = 1
    has = rng.random(n) < 0.33
    t["has_indel"] = has
    pick = rng.integers(0, len(pool), n)
    indel[has] = pool[pick[has]]
    indel["found"][has & (rng.random(n) < 0.2)] = 0                        # Match ran and found nothing
    indel["pos"][has] = t["lev_pos"][has] + rng.integers(-3, 4, int(has.sum()))
    read_ids = (np.cumsum(rng.integers(1, 5, n_reads)) + (7 << 32)).astype(np.uint64)
    return t, indel, read_first, read_ids


def gpu_result_groups(ctx, tasks, indel, read_first, read_ids, id_bytes, max_no_mappings, capacity=None):
    """commit + result-group emission through the _dev entry points on device tensors: (committed, bytes, read offsets)."""
    import torch
    from whisper_amd import api
    dev = torch.device("cuda", 0)

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)

    n, nr = len(tasks), len(read_ids)
    d_t, d_i, d_rf, d_ids = up(tasks), up(indel), up(read_first.astype(np.uint32)), up(read_ids.astype(np.uint64))
    d_c = torch.zeros(max(n, 1) * api.MAPPING.itemsize, dtype=torch.uint8, device=dev)
    cap = capacity or (nr * 16 + n * 11 + 64)
    d_out = torch.full((cap,), 0xEE, dtype=torch.uint8, device=dev)
    d_nb = torch.zeros(1, dtype=torch.int32, device=dev)
    d_off = torch.zeros(nr + 1, dtype=torch.int32, device=dev)
    d_ws = torch.zeros(ctx.result_groups_workspace_bytes(n, nr), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.commit_candidates_dev(n, d_t, d_i, d_c)
    ctx.emit_result_groups_dev(n, d_t, d_c, nr, d_rf, d_ids, api.GroupParams(id_bytes, 4, max_no_mappings, 0), d_out, cap,
                               d_nb, d_off, d_ws)
    ctx.synchronize()
    flags = ctx.take_flags()
    nb = int(d_nb.item())
    return (d_c.cpu().numpy().view(api.MAPPING)[:n], d_out.cpu().numpy()[:min(nb, cap)], d_off.cpu().numpy()[:nr].view(np.uint32),
            flags, d_out.cpu().numpy())


# ---- the suffix-array loops (mapping_core_sa.cpp:193-513) ---------------------------------------------------------
def suffix_array(codes, key_len=64, clean_len=32):
    """Positions of `codes` (numpy, one code per symbol) in the order of their suffixes, as the reference's index holds
    them for one strand; suffixes that meet a non-ACGT symbol within clean_len symbols are left out."""
    n = len(codes)
    bad = (codes > 3).astype(np.int64)
    cs = np.concatenate([[0], np.cumsum(bad)])
    pos = np.arange(0, n - key_len)
    pos = pos[cs[pos + clean_len] - cs[pos] == 0]
    win = np.lib.stride_tricks.sliding_window_view(codes, key_len)[pos]
    order = np.lexsort(win.T[::-1])          # the last key is the primary one
    return pos[order].astype(np.uint32)


def sa_loop_genome(seed, n_symbols=1 << 15, families=48):
    """A small genome with planted near-repeats (families of a 300-symbol segment copied to several places at 2 %
    divergence): anchors then have several suffix-array entries, some of which verify and some of which do not."""
    import torch
    codes = synth.make_genome(n_symbols, seed=seed).numpy().copy()
    rng = np.random.default_rng(seed)
    lo, hi = synth.BOUNDARY_N_LEN + 400, len(codes) - synth.BOUNDARY_N_LEN - 800
    for _ in range(families):
        src = rng.integers(lo, hi)
        seg = codes[src:src + 300].copy()
        seg[seg > 3] = 0
        for _ in range(rng.integers(2, 7)):
            dst = rng.integers(lo, hi)
            s2 = seg.copy()
            mut = rng.random(300) < 0.02
            s2[mut] = (s2[mut] + rng.integers(1, 4, mut.sum())) & 3
            codes[dst:dst + 300] = s2
    t = torch.from_numpy(codes)
    return t, synth.pack_nibbles(t).numpy()


def sa_loop_workload(seed, n_loci, per_locus, m, k, seg, min_read_len=None):
    """Reads for the suffix-array loops: n_loci loci on either strand, per_locus reads each (same anchor: the second and
    later ones take search_in_known_range), 0 .. k + 2 edits outside the anchor segment `seg` (substitutions, now and
    then a short indel), reads of min_read_len .. min_read_len + 3 symbols, sorted by anchor as the bins deliver them.
    Returns (codes, packed, reads (blob, off, lens), sa_dir, sa_rc)."""
    import torch
    codes_t, packed = sa_loop_genome(seed)
    codes = codes_t.numpy()
    G = len(codes)
    comp = np.array([3, 2, 1, 0, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], np.uint8)
    rc_codes = comp[codes[::-1]]
    sa_dir, sa_rc = suffix_array(codes), suffix_array(rc_codes)
    rng = np.random.default_rng(seed + 1)
    fl, fr = seg
    min_read_len = min_read_len or m
    reads = []
    for _ in range(n_loci):
        strand = rng.integers(0, 2)
        text = rc_codes if strand else codes
        while True:
            p = int(rng.integers(synth.BOUNDARY_N_LEN + 60, G - synth.BOUNDARY_N_LEN - m - 80))
            if text[p - 20:p + m + 30].max() <= 3:
                break
        for _ in range(per_locus):
            mi = min_read_len + int(rng.integers(0, 4))
            r = list(text[p:p + mi + 12])
            for _ in range(int(rng.integers(0, k + 3))):
                z = int(rng.integers(0, mi))
                if fl <= z < fr:
                    continue
                u = rng.random()
                if u < 0.8:
                    r[z] = (r[z] + int(rng.integers(1, 4))) & 3
                elif u < 0.9 and z > fr:
                    del r[z]
                elif z > fr:
                    r.insert(z, int(rng.integers(0, 4)))
            reads.append(np.array(r[:mi], np.uint8))
    reads.sort(key=lambda r: bytes(r[fl:fr]))
    blob, off, lens = mixed_reads(reads, 0)
    return codes_t, packed, (blob, off, lens), sa_dir, sa_rc


def sa_loop_tasks(ranges, lens, sa_dir, sa_rc, ref_size, seg, min_read_len, k, max_indel):
    """One task of every kind per suffix-array entry the loops verified (ranges: n_reads x 4 = dir start, dir last, rc
    start, rc last), in the loops' order: a read's direct-strand entries, then its reverse-strand ones.  Reads are
    uploaded twice - n reads as they are, then their reverse complements: CEditDist::LevDiag takes the reverse read
    itself on the other strand (mapping_core_sa.cpp:291), the other verifiers the read in rev_comp orientation.
    Returns (cf, dist, lev, indel, commit skeleton, read_first)."""
    from oracle import bindings as ob
    fl, fr = seg
    n = len(lens)
    rows = []
    read_first = [0]
    for r in range(n):
        for strand in (0, 1):
            s, e = int(ranges[r, 2 * strand]), int(ranges[r, 2 * strand + 1])
            sa = sa_rc if strand else sa_dir
            rows += [(r, strand, int(sa[i])) for i in range(s, e)]
        read_first.append(len(rows))
    c = len(rows)
    rd = np.array([x[0] for x in rows], np.int64)
    st = np.array([x[1] for x in rows], np.int64)
    sp = np.array([x[2] for x in rows], np.int64)
    m = lens[rd].astype(np.int64) if c else np.zeros(0, np.int64)
    direct = st == 0
    cf = np.zeros(c, ob.CF_TASK)
    cf["read_idx"] = rd
    cf["position"] = sp
    cf["read_offset"] = np.where(direct, 0, m - min_read_len)
    cf["pattern_len"] = np.where(m < min_read_len, 0, min_read_len)      # 0 = short read: no filter
    cf["fixed_left"] = np.where(direct, fl, min_read_len - fr)
    cf["fixed_right"] = np.where(direct, fr, min_read_len - fl)
    cf["dir_gen_flag"] = direct
    cf["flags"] = ~direct
    start = np.where(direct, sp - fl, ref_size - sp - m + fl)               # genome symbol under the oriented read's first
    dist = np.zeros(c, ob.DIST_TASK)
    dist["read_idx"] = rd
    dist["ref_pos"] = (start - k).astype(np.uint32)
    dist["max_distance_in_ref"] = m + 2 * k
    dist["max_ed"] = k
    dist["flags"] = ~direct
    lev = np.zeros(c, ob.LEVDIAG_TASK)
    lev["read_idx"] = np.where(direct, rd, n + rd)
    lev["position_in_text"] = sp
    lev["fixed_left"] = np.where(direct, fl, m - fr)
    lev["fixed_right"] = np.where(direct, fr, m - fl)
    lev["max_mismatches"] = k
    lev["dir_gen_flag"] = direct
    indel = np.zeros(c, ob.INDEL_TASK)
    indel["read_idx"] = rd
    indel["ref_pos"] = start.astype(np.uint32)
    indel["max_indel_size"] = max_indel
    indel["fixed_segment_pos"] = np.where(direct, fl, m - fr)
    indel["fixed_segment_size"] = fr - fl
    indel["max_no_mismatches"] = k
    indel["flags"] = ~direct
    commit = np.zeros(c, ob.COMMIT_TASK)
    commit["lev_pos"] = np.where(direct, sp - fl, ref_size - (sp - fl) - 1).astype(np.uint32)
    commit["max_mismatches"] = k
    commit["pattern_len"] = m
    commit["direct"] = direct
    return cf, dist, lev, indel, commit, np.array(read_first, np.uint32)


def with_reverse_reads(reads):
    """(blob, off, lens) of nibble reads followed by their reverse complements (reverse_read, mapping_core.cpp:1063-1108)."""
    import torch
    blob, off, lens = reads
    out = []
    for i in range(len(lens)):
        b = blob[int(off[i]):int(off[i]) + (int(lens[i]) + 1) // 2]
        c = np.zeros(2 * len(b), np.uint8)
        c[0::2], c[1::2] = b >> 4, b & 15
        out.append(c[:int(lens[i])])
    rc = [synth.rev_comp(torch.from_numpy(c.copy())).numpy() for c in out]
    return mixed_reads(out + rc, 0)


# ---- find_pairs<LevMyers>'s chain per hit (sam_matching.cpp:379-457) ---------------------------------------------------
RESCUE_MODEL = (330.0, 12.0)       # mean, deviation: distanceThreshold = (ref_pos_t)(mean + 7 dev) = 414


def rescue_workload(n, seed, m=150, gen=None, frac=0.09):
    """Hits of one read of a pair and the mate the rescue looks for beside them: mates with substitutions, short
    indels, the refinement pattern, junk ends (boundary clipping), too many events (the num_events test), nothing to
    find; hits whose mate lies over the chromosome's end (clipIllegalPositions), in the separator or in the next
    chromosome (Translate's verdict), hits that themselves have too many events.  Returns (genome, mate reads as
    ASCII codes (blob, off, lens), chromosome table, and per task: mate_read_idx, hit_pos, hit_seq_id, hit_num_events,
    paired_dir_direct, max_mate_edit_distance)."""
    import torch
    codes, packed = gen if gen is not None else genome()
    G = codes.numel()
    desc = chromosomes(G)
    g = torch.Generator().manual_seed(seed)
    thr = int(RESCUE_MODEL[0] + 7.0 * RESCUE_MODEL[1])
    k = int(m * frac) * 2

    def rnd(lo, hi):
        return int(torch.randint(lo, hi, (1,), generator=g))

    clean = codes.clone()
    clean[clean > 3] = 0
    reads = []
    hit_pos, seq_id, events, direct = np.zeros(n, np.uint32), np.zeros(n, np.int32), np.zeros(n, np.uint16), np.zeros(n, np.uint8)
    for i in range(n):
        kind = i % 12
        c = rnd(0, 3)
        c_pos, c_size = int(desc[c, 1]), int(desc[c, 0])
        isize = rnd(m + 40, thr - 4)
        mate_direct = bool(rnd(0, 2))
        x = c_pos + rnd(1000, c_size - 2000)              # the mate's first base
        if kind == 9:                                      # over the end of the chromosome: clipped, still this chromosome
            x = c_pos + c_size - m + rnd(1, 40)
        elif kind == 10:                                   # begins in the separator behind it / in the next chromosome
            x = c_pos + c_size + rnd(SEPARATE_N_LEN // 2 - 60, SEPARATE_N_LEN // 2 + 200) if c < 2 else x
        src = clean[x:x + m + 8].clone()
        if len(src) < m + 8:
            src = torch.cat([src, torch.zeros(m + 8 - len(src), dtype=torch.uint8)])
        q = src[:m].clone()
        if kind in (0, 1, 9, 10):
            for _ in range(rnd(0, 6)):
                z = rnd(0, m)
                q[z] = (int(q[z]) + rnd(1, 4)) & 3
        elif kind == 2:                                    # a short deletion / insertion
            z = rnd(10, m - 20)
            q = torch.cat([src[:z], src[z + 2:m + 2]]) if rnd(0, 2) else torch.cat([src[:z], torch.tensor([rnd(0, 4)], dtype=torch.uint8), src[z:m - 1]])
        elif kind == 3:                                    # 1I2M1D: what refine_ext_cigar rewrites
            z = rnd(10, m - 20)
            y = (int(src[z + 3]) + 1) & 3
            q = torch.cat([src[:z], torch.tensor([y], dtype=torch.uint8), src[z:z + 2], src[z + 3:]])[:m]
        elif kind == 4:                                    # junk ends
            cj = rnd(8, 24)
            if rnd(0, 2):
                q[:cj] = torch.randint(0, 4, (cj,), generator=g, dtype=torch.uint8)
            else:
                q[m - cj:] = torch.randint(0, 4, (cj,), generator=g, dtype=torch.uint8)
        elif kind == 5:                                    # many scattered indels: aligned, but too many events
            out, z = [], 0
            for _ in range(k // 2 + 2):
                step = rnd(4, 9)
                out.append(src[z:z + step])
                z += step + 1                              # one deleted base each
            out.append(src[z:])
            q = torch.cat(out)[:m]
        elif kind == 6:                                    # nothing to find
            q = torch.randint(0, 4, (m,), generator=g, dtype=torch.uint8)
        elif kind == 7:                                    # many substitutions: within the distance, few events
            for z in range(5, m, 9):
                q[z] = (int(q[z]) + 1) & 3
        q = q[:m].clone()
        reads.append(q if mate_direct else synth.rev_comp(q))
        direct[i] = mate_direct
        # the hit: leftmost base of its span (what the window formulas of :392-394 assume)
        hit_pos[i] = x + isize - m if mate_direct else x - (isize - m)
        seq_id[i] = c
        events[i] = rnd(0, 4) if kind != 8 else k // 2 + rnd(1, 4)     # kind 8: the hit itself has too many events
    blob, off, lens = mixed_reads(reads, 1)
    return (codes, packed), (blob, off, lens), desc, np.arange(n, dtype=np.uint32), hit_pos, seq_id, events, direct, \
        np.full(n, k, np.uint16)


def rescue_tasks(desc, mate, hit_pos, seq_id, events, direct, k, model=RESCUE_MODEL):
    """wamd_rescue_task as a host would fill it from its chromosome table (ref_desc.cpp:81-118)."""
    from oracle import bindings as ob
    n = len(mate)
    t = np.zeros(n, ob.RESCUE_TASK)
    d = desc.astype(np.int64)
    t["mate_read_idx"] = mate
    t["hit_pos"] = hit_pos
    t["distance_threshold"] = int(model[0] + 7.0 * model[1])
    t["seq_lo"] = d[seq_id, 1] + SEPARATE_N_LEN // 2
    nxt = np.minimum(seq_id + 1, len(d) - 1)
    t["seq_hi"] = np.where(seq_id + 1 < len(d), d[nxt, 1] + SEPARATE_N_LEN // 2, d[seq_id, 1] + d[seq_id, 0] + 2 * SEPARATE_N_LEN)
    t["seq_delta"] = d[seq_id, 2] - d[seq_id, 4] + 1 - d[seq_id, 1]
    t["ref_size"] = d[seq_id, 0] + d[seq_id, 3] + d[seq_id, 2]
    t["max_mate_edit_distance"] = k
    t["hit_num_events"] = events
    t["paired_dir_direct"] = direct
    return t
