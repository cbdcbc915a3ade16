"""Generates tests/golden/example_sam.npz from the reference-held golden output (see tests/example_sam.py):

    python tests/golden/make_example_golden.py        (needs /root/reference; run in the build container)

Inputs: /root/reference/example/out-se.sam, out-pe.sam (the SAMs the reference's CI diffs against) and the two FASTQ
files.  Output: a synthetic genome with the reconstructed reference spans planted between N_ref runs, the reads, one
REC per SAM record with the planted window and the hidden inputs, and the SAM's own POS / CIGAR / MD:Z / NM:i / AS:i.
The hidden inputs (the mapping stage's error count, how often clipBoundaryPositions ran) are chosen by running the
REFERENCE's own objects (oracle/_ref) over the candidates; records it does not reproduce on the planted windows are
kept in the fixture but marked excluded.  Only data is stored: no reference source, no binary.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import example_sam as ex  # noqa: E402
from oracle import bindings  # noqa: E402
from whisper_amd import synth  # noqa: E402

EXAMPLE = "/root/reference/example"
PAD_SMALL, PAD_WIDE, RESCUE_OFFSET = 32, 560, 200
N_PERFECT_PAIRS = 150      # all-perfect pairs kept (the other 1 600 say nothing new)


def main():
    fq = [ex.parse_fastq(os.path.join(EXAMPLE, "SRR10051130_10k_%d.fastq" % i)) for i in (1, 2)]
    se = ex.parse_sam(os.path.join(EXAMPLE, "out-se.sam"))
    pe = ex.parse_sam(os.path.join(EXAMPLE, "out-pe.sam"))
    se_by_name = {r["name"]: r for r in se}

    reads, read_index = [], {}
    genome = [np.full(synth.BOUNDARY_N_LEN, 5, np.uint8)]
    g_len = [synth.BOUNDARY_N_LEN]
    rows, exp_cigar, exp_md, exp_as, src, opts = [], [], [], [], [], []

    def add_read(name, which):
        key = (name, which)
        if key not in read_index:
            read_index[key] = len(reads)
            reads.append(fq[which][name])
        return read_index[key]

    def plant(span, left, right, pad):
        """left / right: codes planted under the soft-clipped ends (N_ref runs, or junk_codes' partial matches)."""
        start = g_len[0]
        block = np.concatenate([np.full(pad, 5, np.uint8), left, ex.ref_codes(span), right, np.full(pad, 5, np.uint8)])
        genome.append(block.astype(np.uint8))
        g_len[0] += len(block)
        return start + pad          # raw position the read's first base (clipped or not) faces

    def n_run(c):
        return np.full(c, 5, np.uint8)

    def junk_codes(seq_part, side):
        """What to plant under a soft-clipped end so that the reference's own clipBoundaryPositions clips exactly there.
        The real flank is unknown; all N_ref would make the end a run of mismatches, and a long such run drags the
        running score below every earlier value: the minimum then lies behind the maximum and the routine leaves the
        record alone (soft_clipping.cpp:892).  So the end is planted as the read's own bases with an N_ref every b + 1
        positions (b matches, one mismatch, ...), b as large as still gives an end that sums below the clipping
        penalty: on the right the first clipped base is the mismatch (the maximum stays on the last kept base), on the
        left the last one (the minimum falls on the last clipped base and nowhere before).  Returns (codes,
        mismatches), or all N_ref when no pattern clips (ends of one or two bases: those come from
        clipOverlappingPairs)."""
        c = len(seq_part)
        own = ex.ref_codes(seq_part)                 # read N -> N_ref: a mismatch either way
        for b in (4, 3, 2, 1, 0):
            codes = own.copy()
            k = np.arange(c)
            mm = (k % (b + 1) == 0) if side == "right" else ((c - 1 - k) % (b + 1) == 0)
            codes[mm] = 5
            bad = codes > 3
            steps = np.where(bad, -5.0, 1.0)
            total = steps.sum()
            if total < -6.0:
                return codes, int(bad.sum())
        return n_run(c), c

    def add_record(r, which, source, span_from=None):
        """One SAM record -> REC row (hidden inputs still open).  span_from: another record of the same read whose
        alignment shows more of the reference (the single-end record of a read whose paired-end record was clipped)."""
        name, flag = r["name"], r["flag"]
        rd = fq[which][name]
        direct = 0 if flag & 16 else 1
        assert r["seq"] == (rd if direct else ex.revcomp(rd)), name
        m = len(rd)
        ops = ex.cigar_ops(r["cigar"])
        c_left = ops[0][0] if ops[0][1] == "S" else 0
        c_right = ops[-1][0] if ops[-1][1] == "S" and len(ops) > 1 else 0
        row = np.zeros(1, ex.REC)[0]
        row["direct"], row["read_idx"], row["mate"] = direct, add_read(name, which), -1
        row["pos"], row["nm"] = r["pos"], r["nm"]
        xm = r["xm"]
        options = []        # (k, n_boundary) candidates, most likely first
        nbs = (0,) if source == "se" else (1, 2, 3)
        seq = r["seq"]
        if xm in ("Perfect", "PerfectDistant"):
            row["kind"] = ex.KIND_PERFECT
            row["coord0"] = r["pos"]
            row["ref_pos"] = plant(ex.reference_span(r["seq"], r["cigar"], r["md"])[0], n_run(0), n_run(0), PAD_SMALL)
            options = [(0, nb) for nb in nbs]
        elif xm in ("FromMapping", "FromMappingDistant", "LevMyers"):
            span, _ = ex.reference_span(r["seq"], r["cigar"], r["md"])
            left, x_left = junk_codes(seq[:c_left], "left") if c_left else (n_run(0), 0)
            right, x_right = junk_codes(seq[m - c_right:], "right") if c_right else (n_run(0), 0)
            base = r["nm"] + x_left + x_right
            if xm == "LevMyers":
                row["kind"] = ex.KIND_RESCUE
                p0 = plant(span, left, right, PAD_WIDE)
                row["ref_pos"], row["W"] = p0 - RESCUE_OFFSET, ex.RESCUE_W
                row["coord0"] = r["pos"] - c_left - RESCUE_OFFSET
                # the call's own budget, k' = 26, unless the planted ends need more
                k = max(ex.RESCUE_K, base + 1)
                options = [(k, nb) for nb in nbs]
            else:
                row["kind"] = ex.KIND_LEV
                p0 = plant(span, left, right, PAD_SMALL)
                row["ref_pos"], row["W"] = p0 - ex.FLANK, m + 2 * ex.FLANK + 1
                row["coord0"] = r["pos"] - c_left - ex.FLANK
                options = [(base + dk, nb) for nb in nbs for dk in (0, -1, 1, 2, 3) if base + dk >= 0]
        elif xm == "Clipping":
            row["kind"] = ex.KIND_SOFTCLIP
            span, _ = ex.reference_span(r["seq"], r["cigar"], r["md"])
            p0 = plant(span, n_run(c_left), n_run(c_right), PAD_WIDE)
            row["ref_pos"], row["W"] = p0 - RESCUE_OFFSET, ex.RESCUE_W
            row["coord0"] = r["pos"] - c_left - RESCUE_OFFSET
            options = [(0, nb) for nb in nbs]
        elif xm == "MappingInsert" and span_from is not None:
            # the record GetExtCigarNew rebuilt: its shape is in the unclipped (single-end) CIGAR aM bI cM
            o2 = ex.cigar_ops(span_from["cigar"])
            assert [o for _, o in o2] == ["M", "I", "M"], span_from["cigar"]
            span, _ = ex.reference_span(span_from["seq"], span_from["cigar"], span_from["md"])
            row["kind"] = ex.KIND_INDEL
            row["indel_type"], row["matching_len1"], row["indel_len1"] = 1, o2[0][0], -o2[1][0]
            p0 = plant(span, n_run(0), n_run(0), PAD_SMALL)
            row["ref_pos"] = p0
            row["coord0"] = span_from["pos"]
            options = [(0, nb) for nb in nbs]
        else:
            return None
        rows.append(row)
        exp_cigar.append(r["cigar"].encode())
        exp_md.append(r["md"].encode())
        exp_as.append(r["as_"].encode())
        src.append(source)
        opts.append(options)
        return len(rows) - 1

    # ---- single-end run: fill_cigar_with_lev / _indel_matching, clipIllegalPositions, no boundary clipping
    skipped = {"se": 0, "pe": 0}
    for r in se:
        if r["flag"] & 4:
            continue
        sf = r if r["xm"] == "MappingInsert" else None
        if add_record(r, 0, "se", sf) is None:
            skipped["se"] += 1
    # ---- paired-end run
    by_pair = {}
    for r in pe:
        by_pair.setdefault(r["name"], []).append(r)
    n_perfect = 0
    for name, pr in by_pair.items():
        if len(pr) != 2 or (pr[0]["flag"] & 4) or (pr[1]["flag"] & 4):
            continue
        if all(x["xm"] == "Perfect" for x in pr):
            n_perfect += 1
            if n_perfect > N_PERFECT_PAIRS:
                continue
        idx = []
        for x in pr:
            which = 0 if x["flag"] & 0x40 else 1
            sf = None
            if x["xm"] == "MappingInsert" and which == 0 and name in se_by_name and \
                    se_by_name[name]["xm"] == "MappingInsert" and (se_by_name[name]["flag"] & 16) == (x["flag"] & 16):
                sf = se_by_name[name]
            i = add_record(x, which, "pe", sf)
            if i is None:
                skipped["pe"] += 1
            idx.append(i)
        if idx[0] is not None and idx[1] is not None:
            rows[idx[0]]["mate"], rows[idx[1]]["mate"] = idx[1], idx[0]
            rows[idx[0]]["first"] = 1 if pr[0]["flag"] & 0x40 else 0
            rows[idx[1]]["first"] = 1 - rows[idx[0]]["first"]

    genome.append(np.full(synth.BOUNDARY_N_LEN, 5, np.uint8))
    import torch
    codes = torch.from_numpy(np.concatenate(genome))
    packed = synth.pack_nibbles(codes).numpy()
    ref_size = codes.numel()
    recs = np.array(rows, ex.REC)
    blob = np.frombuffer("".join(reads).encode(), np.uint8).copy()
    lens = np.array([len(s) for s in reads], np.uint32)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    print("records", len(recs), "reads", len(reads), "genome symbols", ref_size, "skipped", skipped)

    # ---- hidden inputs: the first candidate under which the REFERENCE's own objects give the SAM's strings
    ref = ex.OracleBackend(bindings.Reference(), packed, ref_size, True)
    ref.set_reads(blob, off, lens)
    chosen = np.zeros(len(recs), bool)
    depth = max(len(o) for o in opts)
    for d in range(depth):
        todo = np.nonzero(~chosen & np.array([len(o) > d for o in opts]))[0]
        if not len(todo):
            break
        trial = recs[todo].copy()
        for j, i in enumerate(todo):
            k, nb = opts[i][d]
            if trial["kind"][j] in (ex.KIND_LEV, ex.KIND_RESCUE):
                trial["k"][j] = k
            trial["n_boundary"][j] = nb
        trial["mate"] = -1
        same = ex.compare(trial, [exp_cigar[i] for i in todo], [exp_md[i] for i in todo], [exp_as[i] for i in todo],
                          ex.run_chain(ref, trial, lens, pairs=False))
        for j, i in enumerate(todo):
            if same[j]:
                recs["k"][i], recs["n_boundary"][i] = trial["k"][j], trial["n_boundary"][j]
                chosen[i] = True
        print("candidate", d, "tried", len(todo), "reproduced", int(same.sum()))
    for i in np.nonzero(~chosen)[0]:     # keep a defined input for the excluded ones
        k, nb = opts[i][0]
        recs["k"][i], recs["n_boundary"][i] = (k if recs["kind"][i] in (ex.KIND_LEV, ex.KIND_RESCUE) else 0), nb
    recs["excluded"] = ~chosen
    # pairs: mates of excluded records lose the pair link; then the whole chain with clipOverlappingPairs must still agree
    for i in range(len(recs)):
        mt = recs["mate"][i]
        if mt >= 0 and (recs["excluded"][i] or recs["excluded"][mt]):
            recs["mate"][i] = -1
            recs["mate"][mt] = -1
    same = ex.compare(recs, exp_cigar, exp_md, exp_as, ex.run_chain(ref, recs, lens, pairs=True))
    broke = np.nonzero(~same & (recs["excluded"] == 0))[0]
    for i in broke:
        recs["excluded"][i] = 1
        mt = recs["mate"][i]
        if mt >= 0:
            recs["mate"][mt] = -1
            recs["mate"][i] = -1
    same = ex.compare(recs, exp_cigar, exp_md, exp_as, ex.run_chain(ref, recs, lens, pairs=True))
    assert same[recs["excluded"] == 0].all()
    # the CPU restatement agrees with the reference on every record, excluded or not
    orc = ex.OracleBackend(bindings.Oracle(), packed, ref_size, False)
    orc.set_reads(blob, off, lens)
    a = ex.run_chain(ref, recs, lens, pairs=True)
    b = ex.run_chain(orc, recs, lens, pairs=True)
    for f in a[0].dtype.names:
        if f not in ("no_clippings", "reserved"):      # no_clippings is the C-ABI's bookkeeping, not a field of mapping_desc_t
            assert (a[0][f] == b[0][f]).all(), f
    assert a[1] == b[1] and a[2] == b[2] and (a[3] == b[3]).all()

    kinds = ["PERFECT", "LEV", "RESCUE", "SOFTCLIP", "INDEL"]
    srcs = np.array(src)
    for s in ("se", "pe"):
        for k in range(5):
            sel = (srcs == s) & (recs["kind"] == k)
            if sel.any():
                print(s, kinds[k], "records", int(sel.sum()), "excluded", int(recs["excluded"][sel].sum()))
    clipped = np.array([b"S" in c for c in exp_cigar])
    print("soft-clipped records", int(clipped.sum()), "of them reproduced", int((clipped & (recs["excluded"] == 0)).sum()),
          "; pairs kept for clipOverlappingPairs", int((recs["mate"] >= 0).sum()) // 2)
    w = max(len(c) for c in exp_cigar + exp_md) + 1

    def pad(lst):
        out = np.zeros((len(lst), w), np.uint8)
        for i, s in enumerate(lst):
            out[i, :len(s)] = np.frombuffer(s, np.uint8)
        return out

    np.savez_compressed(os.path.join(HERE, "example_sam.npz"), packed=packed, ref_size=np.uint32(ref_size), blob=blob,
                        off=off, lens=lens, recs=recs, cigar=pad(exp_cigar), md=pad(exp_md), as_=pad(exp_as),
                        source=np.array([0 if s == "se" else 1 for s in src], np.uint8))
    print("wrote example_sam.npz:", os.path.getsize(os.path.join(HERE, "example_sam.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
