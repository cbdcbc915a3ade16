"""Seeded synthetic genomes, reads and verification tasks.

Everything is generated with torch so that the same code runs on the CPU (tests,
small) and on the GPU (bench.py, 2^24 tasks).  Symbols use the reference's codes:
A 0, C 1, G 2, T 3, N_read 4, N_ref 5 (common/defs.h:55-60); packed sequences put
the even position in the HIGH nibble (index/ref_seq.cpp:248-281).
"""
import numpy as np
import torch

BOUNDARY_N_LEN = 10240  # common/defs.h:42

DIST_TASK = np.dtype([("read_idx", "<u4"), ("ref_pos", "<u4"), ("max_distance_in_ref", "<u4"),
                      ("max_ed", "<u2"), ("variant", "u1"), ("flags", "u1")])
LEVDIAG_TASK = np.dtype([("read_idx", "<u4"), ("position_in_text", "<u4"), ("fixed_left", "<u2"),
                         ("fixed_right", "<u2"), ("max_mismatches", "<u2"),
                         ("dir_gen_flag", "<u2")])


def _gen(seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


def make_genome(n_symbols, seed=0xC0FFEE, device="cpu", n_frac=0.001, n_run=50):
    """i.i.d. ACGT with sparse N_ref runs and the reference's N borders.
    Returns one uint8 code per symbol (length n_symbols + 2 * BOUNDARY_N_LEN)."""
    g = _gen(seed, device)
    codes = torch.randint(0, 4, (n_symbols,), generator=g, device=device, dtype=torch.uint8)
    n_runs = int(n_symbols * n_frac / n_run)
    if n_runs > 0 and n_symbols > n_run:
        starts = torch.randint(0, n_symbols - n_run, (n_runs,), generator=g, device=device)
        idx = (starts[:, None] + torch.arange(n_run, device=device)[None, :]).reshape(-1)
        codes[idx] = 5
    border = torch.full((BOUNDARY_N_LEN,), 5, dtype=torch.uint8, device=device)
    return torch.cat([border, codes, border])


def pack_nibbles(codes):
    """[..., n] codes -> [..., ceil(n/2)] bytes, even position in the high nibble."""
    if codes.shape[-1] & 1:
        pad = torch.zeros(codes.shape[:-1] + (1,), dtype=codes.dtype, device=codes.device)
        codes = torch.cat([codes, pad], dim=-1)
    return ((codes[..., 0::2] << 4) | codes[..., 1::2]).contiguous()


def to_ascii(codes):
    lut = torch.tensor([ord(c) for c in "ACGTNNXXXXXXXXXX"], dtype=torch.uint8,
                       device=codes.device)
    return lut[codes.long()]


def rev_comp(codes):
    comp = torch.tensor([3, 2, 1, 0, 4, 5, 6, 7], dtype=torch.uint8, device=codes.device)
    return comp[codes.flip(-1).long()]


def make_reads(genome, n, m, k, seed, true_frac=0.5, rc_frac=0.5, n_read_frac=0.005,
               protect=None, chunk=1 << 15):
    """n reads of length m with their candidate loci.

    "true" reads copy genome[p:p+m] and receive e ~ U{0..k+3} edits (80 % substitutions,
    10 % insertions, 10 % deletions); the others are unrelated random reads.  A fraction
    rc_frac is stored reverse-complemented (the task then carries the rev_comp flag).
    protect = (lo, hi): symbols [lo, hi) of the read stay exact copies of the genome
    (the LevDiag anchor) - also for the unrelated reads.
    Returns dict(codes [n, m] uint8 as stored, p [n] int64 locus start, is_true, is_rc).
    """
    dev = genome.device
    g = _gen(seed, dev)
    G = genome.numel()
    E = k + 3
    lo = BOUNDARY_N_LEN
    hi = G - BOUNDARY_N_LEN - m - E - 8
    out = torch.empty((n, m), dtype=torch.uint8, device=dev)
    p_all = torch.randint(lo, hi, (n,), generator=g, device=dev)
    is_true = torch.rand(n, generator=g, device=dev) < true_frac
    is_rc = torch.rand(n, generator=g, device=dev) < rc_frac
    ar = torch.arange(m, device=dev)
    for b in range(0, n, chunk):
        e_ = min(n, b + chunk)
        c = e_ - b
        p = p_all[b:e_]
        n_edits = torch.randint(0, E + 1, (c,), generator=g, device=dev)
        pos = torch.randint(0, m, (c, E), generator=g, device=dev)
        typ = torch.rand((c, E), generator=g, device=dev)
        active = torch.arange(E, device=dev)[None, :] < n_edits[:, None]
        if protect is not None:
            active &= ~((pos >= protect[0]) & (pos < protect[1]))
        is_sub = active & (typ < 0.8)
        is_ins = active & (typ >= 0.8) & (typ < 0.9)
        is_del = active & (typ >= 0.9)
        # source index: deletions at or before i shift right, insertions before i shift left
        dcnt = (is_del[:, None, :] & (pos[:, None, :] <= ar[None, :, None])).sum(-1)
        icnt = (is_ins[:, None, :] & (pos[:, None, :] < ar[None, :, None])).sum(-1)
        src = p[:, None] + ar[None, :] + dcnt - icnt
        if protect is not None:
            inside = (ar >= protect[0]) & (ar < protect[1])
            src = torch.where(inside[None, :], p[:, None] + ar[None, :], src)
        r = genome[src]
        sub_here = (is_sub[:, None, :] & (pos[:, None, :] == ar[None, :, None])).any(-1)
        ins_here = (is_ins[:, None, :] & (pos[:, None, :] == ar[None, :, None])).any(-1)
        delta = torch.randint(1, 4, (c, m), generator=g, device=dev, dtype=torch.uint8)
        rnd = torch.randint(0, 4, (c, m), generator=g, device=dev, dtype=torch.uint8)
        acgt = r < 4
        r = torch.where(sub_here & acgt, (r + delta) & 3, r)
        r = torch.where(ins_here, rnd, r)
        # unrelated reads
        unrelated = torch.randint(0, 4, (c, m), generator=g, device=dev, dtype=torch.uint8)
        if protect is not None:
            unrelated = torch.where(inside[None, :], genome[p[:, None] + ar[None, :]], unrelated)
        r = torch.where(is_true[b:e_, None], r, unrelated)
        # genome N (5) is not a read symbol: reads carry N_read (4)
        r = torch.where(r > 3, torch.full_like(r, 4), r)
        # a few reads get one N
        has_n = torch.rand(c, generator=g, device=dev) < n_read_frac
        npos = torch.randint(0, m, (c,), generator=g, device=dev)
        if protect is not None:
            has_n &= ~((npos >= protect[0]) & (npos < protect[1]))
        r = torch.where(has_n[:, None] & (ar[None, :] == npos[:, None]), torch.full_like(r, 4), r)
        r = torch.where(is_rc[b:e_, None], rev_comp(r), r)
        out[b:e_] = r
    return dict(codes=out, p=p_all, is_true=is_true, is_rc=is_rc)


def pack_reads(codes, encoding):
    """[n, m] codes -> (blob uint8 [n * stride], offsets uint64 [n + 1], lens uint32 [n])."""
    n, m = codes.shape
    blob = pack_nibbles(codes) if encoding == 0 else to_ascii(codes)
    stride = blob.shape[1]
    offsets = torch.arange(n + 1, device=codes.device, dtype=torch.int64) * stride
    lens = torch.full((n,), m, dtype=torch.int32, device=codes.device)
    return blob.reshape(-1).contiguous(), offsets, lens


def distance_tasks(reads, m, k, W=None, variant=0):
    """One dynamicProgramming task per read: window start p - k, W = m + 2k
    (mapping_core_sa.cpp:237) unless given."""
    n = reads["p"].numel()
    t = np.zeros(n, DIST_TASK)
    t["read_idx"] = np.arange(n, dtype=np.uint32)
    t["ref_pos"] = (reads["p"] - k).cpu().numpy().astype(np.uint32)
    t["max_distance_in_ref"] = (m + 2 * k) if W is None else W
    t["max_ed"] = k
    t["variant"] = variant
    t["flags"] = reads["is_rc"].cpu().numpy().astype(np.uint8)
    return t


def levdiag_tasks(reads, k, fixed_left, fixed_right, cur_offset=0):
    """One LevDiag task per read, direct genome (edit_dist.cpp:112-113):
    window start = position_in_text - cur_offset - k = p - k."""
    n = reads["p"].numel()
    t = np.zeros(n, LEVDIAG_TASK)
    t["read_idx"] = np.arange(n, dtype=np.uint32)
    t["position_in_text"] = (reads["p"] + cur_offset).cpu().numpy().astype(np.uint32)
    t["fixed_left"] = fixed_left
    t["fixed_right"] = fixed_right
    t["max_mismatches"] = k
    t["dir_gen_flag"] = 1
    return t


def stage_segments(min_len, stage_major):
    """adjust_stage_segments, mapper.cpp:616: segment j of stage i."""
    i = stage_major
    return [((min_len * j + i) // (i + 1), (min_len * (j + 1) + i) // (i + 1)) for j in range(i + 1)]


def indel_match_workload(genome, n, m, seed, max_indel=50, max_mm=12, stage_major=5):
    """Shape S8, vectorised: candidates for CIndelMatching::Match.  A quarter each:
    one long deletion (3..max_indel) left of the anchor segment, one right of it, a plain read
    with a few substitutions, and a read that shares only the anchor.  Returns the read codes
    [n, m] and the INDEL_TASK fields as a dict of numpy arrays."""
    dev = genome.device
    g = _gen(seed, dev)
    G = genome.numel()
    segs = torch.tensor(stage_segments(m, stage_major), device=dev)
    sj = torch.randint(0, segs.shape[0], (n,), generator=g, device=dev)
    fl, fr = segs[sj, 0], segs[sj, 1]
    p = torch.randint(BOUNDARY_N_LEN + 200, G - BOUNDARY_N_LEN - m - 200, (n,), generator=g, device=dev)
    kind = torch.arange(n, device=dev) & 3
    L = torch.randint(3, max_indel + 1, (n,), generator=g, device=dev)
    ar = torch.arange(m, device=dev)
    out = torch.empty((n, m), dtype=torch.uint8, device=dev)
    chunk = 1 << 16
    for b in range(0, n, chunk):
        e = min(n, b + chunk)
        c = e - b
        flc, frc, pc, kc, Lc = fl[b:e], fr[b:e], p[b:e], kind[b:e], L[b:e]
        u = torch.rand(c, generator=g, device=dev)
        xl = 8 + (u * (flc - 8).clamp(min=1)).long()            # split left of the anchor
        xr = frc + (u * (m - 8 - frc).clamp(min=1)).long()      # split right of it
        left_ok = (kc == 0) & (flc > 12)
        right_ok = (kc == 1) & (m - frc > 12)
        shift = torch.where(left_ok[:, None] & (ar[None, :] < xl[:, None]), -Lc[:, None], torch.zeros_like(ar)[None, :])
        shift = torch.where(right_ok[:, None] & (ar[None, :] >= xr[:, None]), Lc[:, None], shift)
        r = genome[pc[:, None] + ar[None, :] + shift]
        inside = (ar[None, :] >= flc[:, None]) & (ar[None, :] < frc[:, None])
        rnd = torch.randint(0, 4, (c, m), generator=g, device=dev, dtype=torch.uint8)
        r = torch.where((kc == 3)[:, None] & ~inside, rnd, r)
        sub = (torch.rand((c, m), generator=g, device=dev) < (2.0 / m)) & ~inside
        delta = torch.randint(1, 4, (c, m), generator=g, device=dev, dtype=torch.uint8)
        r = torch.where(sub & (r < 4), (r + delta) & 3, r)
        r = torch.where(r > 3, torch.full_like(r, 4), r)
        out[b:e] = r
    fields = dict(read_idx=np.arange(n, dtype=np.uint32), ref_pos=p.cpu().numpy().astype(np.uint32),
                  max_indel_size=np.full(n, max_indel, np.uint16),
                  fixed_segment_pos=fl.cpu().numpy().astype(np.uint16),
                  fixed_segment_size=(fr - fl).cpu().numpy().astype(np.uint16),
                  max_no_mismatches=np.full(n, max_mm, np.uint8), flags=np.zeros(n, np.uint8))
    return out, fields


def softclip_workload(genome, n, m, W, seed):
    """Mate-rescue shape, vectorised: windows of W symbols holding, a third each,
    a read whose first symbols are junk (-> clipping), a read with one long deletion (3..50), and
    nothing related.  Returns the read codes [n, m] and the wamd_softclip_task fields."""
    dev = genome.device
    g = _gen(seed, dev)
    G = genome.numel()
    p = torch.randint(BOUNDARY_N_LEN + W + 200, G - BOUNDARY_N_LEN - 2 * W - 400, (n,), generator=g, device=dev)
    kind = torch.arange(n, device=dev) % 3
    L = torch.randint(3, 51, (n,), generator=g, device=dev)
    x = torch.randint(25, m - 25, (n,), generator=g, device=dev)
    c = torch.randint(1, 60, (n,), generator=g, device=dev)
    ar = torch.arange(m, device=dev)
    out = torch.empty((n, m), dtype=torch.uint8, device=dev)
    chunk = 1 << 16
    for b in range(0, n, chunk):
        e = min(n, b + chunk)
        k = kind[b:e]
        shift = torch.where((k == 1)[:, None] & (ar[None, :] >= x[b:e, None]), L[b:e, None], torch.zeros_like(ar)[None, :])
        r = genome[p[b:e, None] + ar[None, :] + shift]
        rnd = torch.randint(0, 4, (e - b, m), generator=g, device=dev, dtype=torch.uint8)
        junk = ((k == 0)[:, None] & (ar[None, :] < c[b:e, None])) | (k == 2)[:, None]
        r = torch.where(junk, rnd, r)
        out[b:e] = torch.where(r > 3, torch.full_like(r, 4), r)
    start = p - torch.randint(0, W - m - 60, (n,), generator=g, device=dev)
    fields = dict(read_idx=np.arange(n, dtype=np.uint32), ref_pos=start.cpu().numpy().astype(np.uint32),
                  max_distance_in_ref=np.full(n, W, np.uint32), min_exact_len=np.full(n, 20, np.uint16),
                  flags=np.zeros(n, np.uint8))
    return out, fields
