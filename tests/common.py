"""Shared workload builders for the parity tests (seeded, small)."""
import numpy as np

from whisper_amd import synth

_GENOMES = {}


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


def levdiag_workload(m, k, fl, fr, n, seed, cur_offset=0):
    codes, packed = genome()
    rd = synth.make_reads(codes, n, m, k, seed, rc_frac=0.0, protect=(fl, fr))
    blob, off, lens = host_reads(rd, 0)
    tasks = synth.levdiag_tasks(rd, k, fl, fr, cur_offset)
    return codes, packed, (blob, off, lens), tasks


def stage_segments(min_len, stage_major):
    """adjust_stage_segments, mapper.cpp:616: segment j of stage i."""
    i = stage_major
    return [((min_len * j + i) // (i + 1), (min_len * (j + 1) + i) // (i + 1)) for j in range(i + 1)]


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
