"""The reference-held golden output, example/out-pe.sam / out-se.sam over example/SRR10051130_10k_{1,2}.fastq
(BASELINE configs[0]; compared with `diff` by the reference's CI, .github/workflows/main.yml:29-45), as parity
vectors for the verification path.

The E. coli FASTA the SAMs were made from is absent, but every mapped record spells out the reference under its
alignment: SEQ + CIGAR + MD:Z give the aligned span base by base.  tests/golden/make_example_golden.py plants those
spans in a synthetic genome between N_ref runs (code 5 matches nothing, read N is code 4: a flank cannot take part
in an alignment) and stores, per record, the read as the FASTQ holds it, the strand, the planted window and the SAM's
own POS, CIGAR, MD:Z, NM:i and AS:i strings.  This module holds what the generator and the tests share: parsing,
span reconstruction and `run_chain`, which pushes a fixture's records through the reference's post-processing order
on any backend (the reference's objects, the CPU restatement, the HIP library):

  kind LEV      fill_cigar_with_lev            sam_matching.cpp:689-752   window = read -+ flank
  kind RESCUE   find_pairs<LevMyers>           sam_matching.cpp:390-457   dynamicProgramming over a mate-rescue window,
                                                                          getExtCigar, refine_ext_cigar
  kind SOFTCLIP find_pairs<Clipping>           sam_matching.cpp:543-597   CSoftClipping::match + getExtCigar
  kind INDEL    fill_cigar_with_indel_matching sam_matching.cpp:824-852   GetExtCigarNew
  kind PERFECT  fill_cigar's quick path        sam_matching.cpp:670-680
  then clipIllegalPositions, clipBoundaryPositions (n_boundary times: the paired-end code applies it once per rescue
  attempt that touched the hit and once more at sam.cpp:1890-1892; the single-end code never, sam.cpp:1334-1339),
  clipOverlappingPairs for records whose mate is in the fixture, extract_cigar_and_mdz.

What the SAM does not show and the generator has to choose ("hidden inputs", stored in the fixture): the error count
the mapping stage recorded for a lev record (any k >= the window's distance gives the same alignment; NM:i is k (+1
when refined) minus the edits a clipping removed), how often clipBoundaryPositions ran, and what lay under a
soft-clipped end (N_ref here, so every clipped base is a mismatch before clipping - CIGAR, MD:Z and AS:i of what is
kept do not depend on it).  Records the reference's own objects do not reproduce on the planted windows - an
alignment that used the real flank, a tie decided by it - are listed in the fixture as excluded, with the count.
"""
import re

import numpy as np

KIND_PERFECT, KIND_LEV, KIND_RESCUE, KIND_SOFTCLIP, KIND_INDEL = 0, 1, 2, 3, 4
FLANK = 13          # uint32(max_no_errors * sensitivity_factor) + 1 = uint32(5 * 2.5) + 1 at 151 bp (sam_matching.cpp:696)
RESCUE_K = 26       # uint32(151 * 0.09) * 2 (sam_matching.cpp:390)
RESCUE_W = 660      # a trained model's mean + 7 sigma + 2 k' 
REF_SIZE = 4700000  # longer than the E. coli chromosome: nothing hangs over its end

REC = np.dtype([("kind", "u1"), ("direct", "u1"), ("n_boundary", "u1"), ("excluded", "u1"), ("read_idx", "<u4"),
                ("ref_pos", "<u4"),          # raw position of the window start in the planted genome
                ("W", "<u4"), ("k", "<u4"),   # window length and error budget of the verifier call
                ("coord0", "<i4"),           # chromosome coordinate of raw position ref_pos (Translate stands in)
                ("mate", "<i4"),             # record index of the mate for clipOverlappingPairs, -1 = none
                ("first", "u1"), ("pad", "u1", (3,)),   # 1 = p.first of its pair
                ("matching_len1", "<u4"), ("indel_len1", "<i4"), ("indel_type", "<u4"),   # kind INDEL
                ("pos", "<i4"), ("nm", "<u4")])   # the SAM's POS and NM:i


def double2pchar(val, prec=1):
    """SDouble2PChar (common/utils.h:276-286) as AS:i: is printed: the integer part, then the digits of
    int((1 + fraction) * 10^prec + 0.5) with the leading one overwritten by the point (a carry out of the fraction
    is lost, as there)."""
    a = int(val)
    b = int((1.0 + (val - float(a))) * 10 ** prec + 0.5)
    return str(a) + "." + str(b)[1:]


def parse_sam(path):
    out = []
    for ln in open(path):
        if ln.startswith("@"):
            continue
        f = ln.rstrip("\n").split("\t")
        tags = {t[:4]: t[5:] for t in f[11:]}
        out.append(dict(name=f[0], flag=int(f[1]), pos=int(f[3]), cigar=f[5], seq=f[9], nm=int(tags.get("NM:i", 0)),
                        md=tags.get("MD:Z", ""), as_=tags.get("AS:i", ""), xm=tags.get("XM:Z", "")))
    return out


def parse_fastq(path):
    reads = {}
    with open(path) as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline().strip()
            f.readline()
            f.readline()
            reads[h[1:].split()[0]] = s
    return reads


_COMP = str.maketrans("ACGTN", "TGCAN")


def revcomp(s):
    return s.translate(_COMP)[::-1]


def cigar_ops(cigar):
    return [(int(n), o) for n, o in re.findall(r"(\d+)([MIDS])", cigar)]


def reference_span(seq, cigar, md):
    """The reference bases under the aligned part of a record, from SEQ (reference strand), CIGAR and MD:Z; also the
    ext-CIGAR that spells the same alignment ('.' match, base = mismatch, '^'base deletion, '#'base insertion, '$'
    clipped).  MD:Z is the reference's flavour: a count (also 0) in front of every deleted base."""
    toks = re.findall(r"\d+|\^[A-Z]|[A-Z]", md)
    ti, left = 0, 0          # cursor into toks, matches still to consume from the current count

    def next_md():
        nonlocal ti, left
        while True:
            if left > 0:
                left -= 1
                return None
            if ti >= len(toks):
                raise ValueError("MD:Z shorter than the CIGAR")
            t = toks[ti]
            ti += 1
            if t.isdigit():
                left = int(t)
                continue
            return t

    span, ext, q = [], [], 0
    for n, op in cigar_ops(cigar):
        if op == "S":
            ext.append("$" * n)
            q += n
        elif op == "I":
            ext.append("".join("#" + c for c in seq[q:q + n]))
            q += n
        elif op == "M":
            for _ in range(n):
                t = next_md()
                if t is None:
                    span.append(seq[q])
                    ext.append(".")
                else:
                    if t.startswith("^"):
                        raise ValueError("deletion inside an M run")
                    span.append(t)
                    ext.append(t)
                q += 1
        elif op == "D":
            for _ in range(n):
                t = next_md()
                if t is None or not t.startswith("^"):
                    raise ValueError("D without ^ in MD:Z")
                span.append(t[1])
                ext.append(t)
    if q != len(seq):
        raise ValueError("CIGAR does not cover SEQ")
    return "".join(span), "".join(ext)


_CODE = {"A": 0, "C": 1, "G": 2, "T": 3}


def ref_codes(s):
    return np.array([_CODE.get(c, 5) for c in s], np.uint8)     # anything else in a reference: N_ref


# ---- backends -------------------------------------------------------------------------------------------------
class OracleBackend:
    """oracle.bindings.Oracle or .Reference behind one interface (both take host arrays)."""

    def __init__(self, lib, genome, ref_size, is_reference):
        self.lib, self.genome, self.ref_size, self.is_ref = lib, genome, ref_size, is_reference

    def set_reads(self, blob, off, lens):
        self.reads = (blob, off, lens)

    def align(self, tasks, stride):
        b, o, l = self.reads
        if self.is_ref:
            rc, res, cig = self.lib.align(self.genome, self.ref_size, b, o, l, 1, tasks, 0, 1, cigar_stride=stride)
        else:
            rc, res, cig = self.lib.align(self.genome, b, o, l, 1, tasks, 0, 1, cigar_stride=stride)
        assert rc == 0
        return res, cig

    def softclip(self, tasks, stride):
        from oracle import bindings as ob
        b, o, l = self.reads
        rc, res, cig = self.lib.softclip_match(self.genome, b, o, l, 1, tasks, ob.default_softclip_params(fresh_object=0),
                                               cigar_stride=stride)
        assert rc == 0
        return res, cig

    def indel_ext_cigar(self, tasks, stride):
        b, o, l = self.reads
        rc, res, cig = self.lib.indel_ext_cigar(self.genome, b, o, l, 1, tasks, cigar_stride=stride)
        assert rc == 0
        return res, cig

    def clip_illegal(self, t, rec, cig):
        rc, r, c = self.lib.clip_illegal(t, rec, cig)
        assert rc == 0
        return r, c

    def clip_boundary(self, rec, cig):
        rc, r, c = self.lib.clip_boundary(rec, cig)
        assert rc == 0
        return r, c

    def clip_pairs(self, t, rec, cig):
        rc, r, c = self.lib.clip_overlapping_pairs(t, rec, cig)
        assert rc == 0
        return r, c

    def sam_cigar(self, ext):
        rc, c, m, b, l = self.lib.sam_cigar(ext)
        assert rc == 0
        return c, m, l


class GpuBackend:
    """whisper_amd.api.Context: every stage through the C-ABI."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.reads = None

    def set_reads(self, blob, off, lens):
        from whisper_amd import api
        if self.reads is not None:
            self.reads.free()
        self.reads = self.ctx.upload_reads(blob, off, lens, api.READS_ASCII, orientations=3)

    def align(self, tasks, stride):
        return self.ctx.align(self.reads, tasks, 0, 1, cigar_stride=stride)

    def softclip(self, tasks, stride):
        from whisper_amd import api
        return self.ctx.softclip_match(self.reads, tasks, api.default_softclip_params(fresh_object=0), cigar_stride=stride)

    def indel_ext_cigar(self, tasks, stride):
        return self.ctx.indel_ext_cigar(self.reads, tasks, cigar_stride=stride)

    def clip_illegal(self, t, rec, cig):
        return self.ctx.clip_illegal(t, rec, cig)

    def clip_boundary(self, rec, cig):
        return self.ctx.clip_boundary(rec, cig)

    def clip_pairs(self, t, rec, cig):
        return self.ctx.clip_overlapping_pairs(t, rec, cig)

    def sam_cigar(self, ext):
        c, m, b, l = self.ctx.sam_cigar(ext)
        return c, m, l

    def close(self):
        if self.reads is not None:
            self.reads.free()
            self.reads = None


# ---- the chain ------------------------------------------------------------------------------------------------
def run_chain(be, recs, lens, stride=512, pairs=True):
    """Every record of `recs` (dtype REC) through its chain on backend `be`; returns (clip records, CIGAR strings,
    MD:Z strings, ok mask).  `lens`: read lengths of the uploaded batch."""
    from oracle import bindings as ob      # numpy dtypes of the C-ABI structs only
    n = len(recs)
    rec = np.zeros(n, ob.CLIP_RECORD)
    ext = np.zeros((n, stride), np.uint8)
    ok = np.ones(n, bool)
    perfect = recs["kind"] == KIND_PERFECT
    m_all = lens[recs["read_idx"]].astype(np.int64)
    # --- stage 1 per kind
    rec["score"][perfect] = m_all[perfect] * 1.0                      # sam_matching.cpp:670-680
    rec["ref_length"][perfect] = m_all[perfect]
    rec["ref_seq_pos"][perfect] = recs["coord0"][perfect]
    for kind in (KIND_LEV, KIND_RESCUE):
        idx = np.nonzero(recs["kind"] == kind)[0]
        if not len(idx):
            continue
        t = np.zeros(len(idx), ob.DIST_TASK)
        t["read_idx"] = recs["read_idx"][idx]
        t["ref_pos"] = recs["ref_pos"][idx]
        t["max_distance_in_ref"] = recs["W"][idx]
        t["max_ed"] = recs["k"][idx]
        t["flags"] = 1 - recs["direct"][idx]
        res, cig = be.align(t, stride)
        ok[idx] = res["ok"] == 1
        ext[idx] = cig
        rec["score"][idx] = res["affine_score"]
        rec["ref_seq_pos"][idx] = recs["coord0"][idx] + res["begin_pos"].astype(np.int64)
        rec["ref_length"][idx] = (res["end_pos"] - res["begin_pos"]).astype(np.uint16)
        if kind == KIND_LEV:     # the mapping stage's count stays, +1 when refined (sam_matching.cpp:727-735)
            rec["err_edit_distance"][idx] = recs["k"][idx] + res["refined"]
        else:                    # construct<lev>(..., edit_distance) after refine_ext_cigar (sam_matching.cpp:424-441)
            rec["err_edit_distance"][idx] = res["edit_distance"] + res["refined"]
    idx = np.nonzero(recs["kind"] == KIND_SOFTCLIP)[0]
    if len(idx):
        t = np.zeros(len(idx), ob.SOFTCLIP_TASK)
        t["read_idx"] = recs["read_idx"][idx]
        t["ref_pos"] = recs["ref_pos"][idx]
        t["max_distance_in_ref"] = recs["W"][idx]
        t["min_exact_len"] = 20
        t["flags"] = 1 - recs["direct"][idx]
        res, cig = be.softclip(t, stride)
        ok[idx] = res["found"] == 1
        ext[idx] = cig
        m = m_all[idx]
        rec["score"][idx] = res["affine_score"]                                   # sam_matching.cpp:585-597
        rec["left_clipping"][idx] = res["left_clipping"]
        rec["right_clipping"][idx] = res["right_clipping"]
        rec["ref_length"][idx] = (m - res["left_clipping"] - res["right_clipping"] + res["del_size"]).astype(np.uint16)
        rec["err_edit_distance"][idx] = np.abs(res["del_size"]) + res["no_mismatches"]
        rec["ref_seq_pos"][idx] = recs["coord0"][idx] + res["pos"].astype(np.int32) + res["left_clipping"].astype(np.int64)
    idx = np.nonzero(recs["kind"] == KIND_INDEL)[0]
    if len(idx):
        t = np.zeros(len(idx), ob.INDEL_CIGAR_TASK)
        t["read_idx"] = recs["read_idx"][idx]
        t["examined_start_pos"] = recs["ref_pos"][idx]
        t["type"] = recs["indel_type"][idx]
        t["matching_len1"] = recs["matching_len1"][idx]
        t["indel_len1"] = recs["indel_len1"][idx]
        t["flags"] = 1 - recs["direct"][idx]
        res, cig = be.indel_ext_cigar(t, stride)
        ok[idx] = res["ok"] == 1
        ext[idx] = cig
        rec["score"][idx] = res["score"]
        rec["err_edit_distance"][idx] = res["err_edit_distance"]
        rec["ref_length"][idx] = res["ref_length"]
        rec["left_clipping"][idx] = res["left_clipping"]
        rec["right_clipping"][idx] = res["right_clipping"]
        rec["ref_seq_pos"][idx] = recs["coord0"][idx]
    # records whose verifier gave up keep an inert string (the reference does not survive them, see whisper_amd.h)
    rec["err_edit_distance"][~ok] = 0
    ext[~ok] = 0
    # --- clipIllegalPositions (sam.cpp:1339 / 1889)
    it = np.zeros(n, ob.CLIP_ILLEGAL_TASK)
    it["ref_size"] = REF_SIZE
    it["read_length"] = m_all
    it["perfect"] = perfect
    rec, ext = be.clip_illegal(it, rec, ext)
    # --- clipBoundaryPositions, as often as the host applied it
    for rnd in range(int(recs["n_boundary"].max()) if n else 0):
        idx = np.nonzero(recs["n_boundary"] > rnd)[0]
        r2, e2 = be.clip_boundary(rec[idx], ext[idx])
        rec[idx], ext[idx] = r2, e2
    # --- clipOverlappingPairs (sam.cpp:1898-1900) for pairs with both mates here
    if pairs:
        first = np.nonzero((recs["mate"] >= 0) & (recs["first"] == 1))[0]
        first = np.array([i for i in first if ok[i] and ok[recs["mate"][i]]], np.int64)
        if len(first):
            second = recs["mate"][first]
            order = np.stack([first, second], 1).reshape(-1)
            pt = np.zeros(len(first), ob.CLIP_PAIR_TASK)
            pt["ref_seq_id"] = 0
            pt["read_length"][:, 0] = m_all[first]
            pt["read_length"][:, 1] = m_all[second]
            pt["direct"][:, 0] = recs["direct"][first]
            pt["direct"][:, 1] = recs["direct"][second]
            pt["perfect"][:, 0] = perfect[first]
            pt["perfect"][:, 1] = perfect[second]
            r2, e2 = be.clip_pairs(pt, rec[order], ext[order])
            rec[order], ext[order] = r2, e2
    # --- extract_cigar_and_mdz; perfect matches keep "<len>M" / "<len>" (sam.cpp:1504-1515)
    cg, md, ln = be.sam_cigar(ext)
    cigars, mdz = [], []
    for i in range(n):
        if perfect[i]:
            cigars.append(b"%dM" % m_all[i])
            mdz.append(b"%d" % m_all[i])
        else:
            cigars.append(bytes(cg[i, :ln["cigar_len"][i]]))
            mdz.append(bytes(md[i, :ln["mdz_len"][i]]))
    return rec, cigars, mdz, ok


def compare(recs, expect_cigar, expect_md, expect_as, got):
    """Per record: does the chain's output equal the SAM's POS, CIGAR, MD:Z, NM:i, AS:i?"""
    rec, cigars, mdz, ok = got
    same = np.zeros(len(recs), bool)
    for i in range(len(recs)):
        same[i] = (ok[i] and cigars[i] == expect_cigar[i] and mdz[i] == expect_md[i] and
                   int(rec["err_edit_distance"][i]) == int(recs["nm"][i]) and
                   int(rec["ref_seq_pos"][i]) == int(recs["pos"][i]) and
                   double2pchar(float(rec["score"][i])).encode() == expect_as[i])
    return same
