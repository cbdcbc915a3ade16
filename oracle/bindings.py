"""ctypes bindings for the two CHECKERS: oracle/libwv_oracle.so (our CPU restatement)
and oracle/_ref/libwhisper_ref.so (the reference's own classes behind ref_shim.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under whisper_amd/ imports this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libwv_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libwhisper_ref.so")

# numpy views of the C-ABI structs (include/whisper_amd.h)
DIST_TASK = np.dtype([("read_idx", "<u4"), ("ref_pos", "<u4"), ("max_distance_in_ref", "<u4"),
                      ("max_ed", "<u2"), ("variant", "u1"), ("flags", "u1")])
DIST_RESULT = np.dtype([("edit_distance", "<u4"), ("pos", "<u4")])
LEVDIAG_TASK = np.dtype([("read_idx", "<u4"), ("position_in_text", "<u4"), ("fixed_left", "<u2"),
                         ("fixed_right", "<u2"), ("max_mismatches", "<u2"),
                         ("dir_gen_flag", "<u2")])
ALIGN_RESULT = np.dtype([("affine_score", "<f8"), ("ok", "<u4"), ("edit_distance", "<u4"),
                         ("end_pos", "<u4"), ("begin_pos", "<u4"), ("num_events", "<u4"),
                         ("cigar_len", "<u4"), ("refined", "<u4"), ("reserved", "<u4")])
INDEL_TASK = np.dtype([("read_idx", "<u4"), ("ref_pos", "<u4"), ("max_indel_size", "<u2"),
                       ("fixed_segment_pos", "<u2"), ("fixed_segment_size", "<u2"),
                       ("max_no_mismatches", "u1"), ("flags", "u1")])
MAPPING = np.dtype([("found", "<u4"), ("type", "<u4"), ("pos", "<u4"), ("no_mismatches", "<u4"),
                    ("matching_len1", "<u4"), ("matching_len2", "<u4"), ("indel_len1", "<i4"),
                    ("indel_len2", "<i4"), ("clipping_len", "<u4"), ("penalty", "<u4")])
INDEL_CIGAR_TASK = np.dtype([("read_idx", "<u4"), ("examined_start_pos", "<u4"), ("type", "<u4"),
                             ("matching_len1", "<u4"), ("matching_len2", "<u4"), ("indel_len1", "<i4"),
                             ("indel_len2", "<i4"), ("clipping_len", "<u4"), ("flags", "<u4"),
                             ("reserved", "<u4")])
INDEL_CIGAR_RESULT = np.dtype([("score", "<f8"), ("ok", "<u4"), ("cigar_len", "<u4"),
                               ("err_edit_distance", "<u2"), ("num_events", "<u2"),
                               ("ref_length", "<u2"), ("left_clipping", "<u2"),
                               ("right_clipping", "<u2"), ("method", "<u2"), ("reserved", "<u4")])
CLIP_RECORD = np.dtype([("score", "<f8"), ("ref_seq_pos", "<i4"), ("ref_length", "<u2"),
                        ("err_edit_distance", "<u2"), ("left_clipping", "<u2"),
                        ("right_clipping", "<u2"), ("cigar_len", "<u4"), ("no_clippings", "<u4"),
                        ("reserved", "<u4")])
assert DIST_TASK.itemsize == 16 and LEVDIAG_TASK.itemsize == 16 and ALIGN_RESULT.itemsize == 40
assert INDEL_TASK.itemsize == 16 and MAPPING.itemsize == 40 and INDEL_CIGAR_TASK.itemsize == 40
assert INDEL_CIGAR_RESULT.itemsize == 32 and CLIP_RECORD.itemsize == 32



CF_TASK = np.dtype([("read_idx", "<u4"), ("position", "<u4"), ("read_offset", "<u2"), ("pattern_len", "<u2"),
                    ("fixed_left", "<u2"), ("fixed_right", "<u2"), ("dir_gen_flag", "u1"), ("flags", "u1"),
                    ("reserved", "<u2")])


SAM_CIGAR_LEN = np.dtype([("cigar_len", "<u4"), ("mdz_len", "<u4"), ("cigar_bin_len", "<u4"), ("reserved", "<u4")])


def _sam_cigar(fn, ext_cigars, out_stride=None):
    ext = np.ascontiguousarray(ext_cigars, np.uint8)
    n, es = ext.shape
    os_ = out_stride or 2 * es + 16
    cig = np.zeros((n, os_), np.uint8)
    mdz = np.zeros((n, os_), np.uint8)
    cbin = np.zeros((n, es + 4), np.uint32)
    lens = np.zeros(n, SAM_CIGAR_LEN)
    rc = fn(C.c_uint32(n), _p(ext), C.c_uint32(es), _p(cig), C.c_uint32(os_), _p(mdz), C.c_uint32(os_), _p(cbin),
            C.c_uint32(es + 4), _p(lens))
    return rc, cig, mdz, cbin, lens


FINISH_TASK = np.dtype([("read_idx", "<u4"), ("pos", "<u4"), ("type", "<u4"), ("penalty", "<u4"), ("matching_len1", "<u4"),
                        ("matching_len2", "<u4"), ("indel_len1", "<i4"), ("indel_len2", "<i4"), ("clipping_len", "<u4"),
                        ("ref_seq_id", "<i4"), ("ref_seq_pos", "<i4"), ("ref_size", "<i4"), ("err_edit_distance", "<u2"),
                        ("direct", "u1"), ("distant", "u1"), ("isize_min", "<i4"), ("isize_max", "<i4"), ("flags", "<u4")])
FINISH_RESULT = np.dtype([("score", "<f8"), ("pos", "<u4"), ("ref_seq_pos", "<i4"), ("ref_length", "<u2"),
                          ("err_edit_distance", "<u2"), ("num_events", "<u2"), ("left_clipping", "<u2"),
                          ("right_clipping", "<u2"), ("method", "<u2"), ("ext_cigar_len", "<u4"), ("cigar_len", "<u4"),
                          ("mdz_len", "<u4"), ("ok", "<u4"), ("reserved", "<u4")])
assert FINISH_TASK.itemsize == 64 and FINISH_RESULT.itemsize == 48


COMMIT_TASK = np.dtype([("lev_pos", "<u4"), ("lev_errors", "<u4"), ("max_mismatches", "<u2"), ("pattern_len", "<u2"),
                        ("direct", "u1"), ("has_indel", "u1"), ("reserved", "<u2")])
assert COMMIT_TASK.itemsize == 16


class GroupParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("id_bytes", "mapping_counter_size", "max_no_mappings", "reserved")]


class FinishParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("flank", "enable_short_indel_refinement", "enable_boundary_clipping",
                                          "enable_paired_clipping")]


def default_finish_params(flank=6):
    # mapper/params.h: refinement, boundary and paired clipping on; flank = max_no_errors (+ sensitivity) + 1
    return FinishParams(flank, 1, 1, 1)


class Scoring(C.Structure):  # noqa: E302
    _fields_ = [(n, C.c_double) for n in ("match", "mismatch", "gap_ins_open", "gap_ins_extend",
                                           "gap_del_open", "gap_del_extend", "clipping")]


class IndelParams(C.Structure):
    _fields_ = [("scoring", Scoring), ("min_clipped_factor", C.c_double), ("max_indel_bound", C.c_uint32),
                ("reserved", C.c_uint32)]


def default_indel_params(min_clipped_factor=1.0):
    return IndelParams(default_scoring(), min_clipped_factor, 0, 0)


def default_scoring():
    # mapper/params.h:145-154
    return Scoring(1.0, -5.0, -5.0, -0.4, -5.0, -0.4, -6.0)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dtype=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


class _Lib:
    path = None

    def __init__(self):
        if not os.path.exists(self.path):
            raise FileNotFoundError(self.path + " (run `make -C oracle`)")
        self.lib = C.CDLL(self.path)


class Oracle(_Lib):
    """Our CPU restatement (oracle/wv_oracle.c)."""
    path = ORACLE_SO

    def distance(self, genome, reads, offsets, lens, encoding, tasks, last_symbol=-1,
                 max_text_len=16384):
        tasks = _c(tasks)
        res = np.zeros(len(tasks), DIST_RESULT)
        rc = self.lib.wo_distance_batch(_p(genome), _p(reads), _p(offsets), _p(lens),
                                        C.c_int(encoding), _p(tasks), C.c_uint32(len(tasks)),
                                        C.c_int(last_symbol), C.c_uint32(max_text_len), _p(res))
        return rc, res

    def levdiag(self, genome, ref_size, cur_offset, reads, offsets, lens, tasks):
        tasks = _c(tasks)
        res = np.zeros(len(tasks), np.uint32)
        rc = self.lib.wo_levdiag_batch(_p(genome), C.c_uint32(ref_size), C.c_uint32(cur_offset),
                                       _p(reads), _p(offsets), _p(lens), _p(tasks),
                                       C.c_uint32(len(tasks)), _p(res))
        return rc, res

    def align(self, genome, reads, offsets, lens, encoding, tasks, last_symbol=0, refine=1,
              max_text_len=16384, scoring=None, cigar_stride=1024):
        tasks = _c(tasks)
        sc = scoring or default_scoring()
        res = np.zeros(len(tasks), ALIGN_RESULT)
        cig = np.zeros((len(tasks), cigar_stride), np.uint8)
        rc = self.lib.wo_align_batch(_p(genome), _p(reads), _p(offsets), _p(lens),
                                     C.c_int(encoding), _p(tasks), C.c_uint32(len(tasks)),
                                     C.c_int(last_symbol), C.c_int(refine),
                                     C.c_uint32(max_text_len), C.byref(sc), _p(res), _p(cig),
                                     C.c_uint32(cigar_stride))
        return rc, res, cig


    def cf_filter(self, genome, ref_size, cur_offset, reads, offsets, lens, tasks):
        """CF pre-filter distances (mapping_core_dp.cpp:349-483); nibble-packed reads."""
        tasks = np.ascontiguousarray(tasks, CF_TASK)
        res = np.zeros(len(tasks), np.uint32)
        rc = self.lib.wo_cf_batch(_p(genome), C.c_uint32(ref_size), C.c_uint32(cur_offset), _p(reads), _p(offsets),
                                  _p(lens), _p(tasks), C.c_uint32(len(tasks)), _p(res))
        return rc, res

    def sam_cigar(self, ext_cigars, out_stride=None):
        """ext-CIGAR -> (rc, CIGAR text, MD:Z, BAM ops, lengths), sam.cpp:805-1051."""
        return _sam_cigar(self.lib.wo_sam_cigar_batch, ext_cigars, out_stride)

    def check_ext_cigars(self, genome, reads, offsets, lens, tasks, results, cigars):
        """Independent replay of ext-CIGARs (ASCII reads); returns a failure code per task, 0 = consistent."""
        tasks = _c(tasks)
        results = _c(results)
        cigars = _c(cigars)
        fail = np.zeros(len(tasks), np.uint8)
        self.lib.wo_check_ext_cigar_batch(_p(genome), _p(reads), _p(offsets), _p(lens), _p(tasks),
                                          C.c_uint32(len(tasks)), _p(results), _p(cigars),
                                          C.c_uint32(cigars.shape[1]), _p(fail))
        return fail

    def indel_match(self, genome, reads, offsets, lens, encoding, tasks, params=None):
        tasks = _c(tasks)
        prm = params or default_indel_params()
        res = np.zeros(len(tasks), MAPPING)
        rc = self.lib.wo_indel_match_batch(_p(genome), _p(reads), _p(offsets), _p(lens),
                                           C.c_int(encoding), _p(tasks), C.c_uint32(len(tasks)),
                                           C.byref(prm), _p(res))
        return rc, res

    def indel_ext_cigar(self, genome, reads, offsets, lens, encoding, tasks, scoring=None,
                        cigar_stride=1024):
        tasks = _c(tasks)
        sc = scoring or default_scoring()
        res = np.zeros(len(tasks), INDEL_CIGAR_RESULT)
        cig = np.zeros((len(tasks), cigar_stride), np.uint8)
        rc = self.lib.wo_indel_ext_cigar_batch(_p(genome), _p(reads), _p(offsets), _p(lens),
                                               C.c_int(encoding), _p(tasks), C.c_uint32(len(tasks)),
                                               C.byref(sc), _p(res), _p(cig), C.c_uint32(cigar_stride))
        return rc, res, cig

    def clip_boundary(self, records, cigars, scoring=None):
        sc = scoring or default_scoring()
        rec = np.array(records, CLIP_RECORD, copy=True)
        cig = np.array(cigars, np.uint8, copy=True)
        rc = self.lib.wo_clip_boundary_batch(C.c_uint32(len(rec)), _p(rec), _p(cig),
                                             C.c_uint32(cig.shape[1]), C.byref(sc))
        return rc, rec, cig

    def serialize(self, mappings, direction):
        m = _c(mappings)
        d = _c(direction, np.uint8)
        out = np.zeros((len(m), 12), np.uint8)
        sizes = np.zeros(len(m), np.uint8)
        rc = self.lib.wo_serialize_batch(C.c_uint32(len(m)), _p(m), _p(d), _p(out), _p(sizes))
        return rc, out, sizes


class Reference(_Lib):
    """The reference's own objects (oracle/_ref/libwhisper_ref.so)."""
    path = REF_SO

    def __init__(self):
        super().__init__()
        self.lib.ref_version.restype = C.c_char_p

    def version(self):
        return self.lib.ref_version().decode()

    def distance(self, genome, ref_size, reads, offsets, lens, encoding, tasks, last_symbol=-1,
                 max_query_len=1024, max_text_len=16384, nthreads=1):
        tasks = _c(tasks)
        res = np.zeros(len(tasks), DIST_RESULT)
        rc = self.lib.ref_distance_batch(_p(genome), C.c_uint32(ref_size), _p(reads), _p(offsets),
                                         _p(lens), C.c_int(encoding), _p(tasks),
                                         C.c_uint32(len(tasks)), C.c_int(last_symbol),
                                         C.c_uint32(max_query_len), C.c_uint32(max_text_len),
                                         _p(res), C.c_int(nthreads))
        return rc, res

    def levdiag(self, genome, ref_size, cur_offset, reads, offsets, lens, tasks,
                max_query_len=1024, max_ed=16, nthreads=1):
        tasks = _c(tasks)
        res = np.zeros(len(tasks), np.uint32)
        rc = self.lib.ref_levdiag_batch(_p(genome), C.c_uint32(ref_size), C.c_uint32(cur_offset),
                                        _p(reads), _p(offsets), _p(lens), _p(tasks),
                                        C.c_uint32(len(tasks)), C.c_uint32(max_query_len),
                                        C.c_uint32(max_ed), _p(res), C.c_int(nthreads))
        return rc, res

    def align(self, genome, ref_size, reads, offsets, lens, encoding, tasks, last_symbol=0,
              refine=1, max_query_len=1024, max_text_len=16384, scoring=None, cigar_stride=1024,
              nthreads=1):
        tasks = _c(tasks)
        sc = scoring or default_scoring()
        res = np.zeros(len(tasks), ALIGN_RESULT)
        cig = np.zeros((len(tasks), cigar_stride), np.uint8)
        rc = self.lib.ref_align_batch(_p(genome), C.c_uint32(ref_size), _p(reads), _p(offsets),
                                      _p(lens), C.c_int(encoding), _p(tasks),
                                      C.c_uint32(len(tasks)), C.c_int(last_symbol),
                                      C.c_int(refine), C.c_uint32(max_query_len),
                                      C.c_uint32(max_text_len), C.byref(sc), _p(res), _p(cig),
                                      C.c_uint32(cigar_stride), C.c_int(nthreads))
        return rc, res, cig


def have_reference():
    return os.path.exists(REF_SO)


def _ref_indel_match(self, genome, reads, offsets, lens, encoding, tasks, params=None, nthreads=1):
    tasks = _c(tasks)
    prm = params or default_indel_params()
    res = np.zeros(len(tasks), MAPPING)
    rc = self.lib.ref_indel_match_batch(_p(genome), _p(reads), _p(offsets), _p(lens), C.c_int(encoding),
                                        _p(tasks), C.c_uint32(len(tasks)), C.byref(prm), _p(res),
                                        C.c_int(nthreads))
    return rc, res


def _ref_indel_ext_cigar(self, genome, reads, offsets, lens, encoding, tasks, max_indel=50, scoring=None,
                         cigar_stride=1024):
    tasks = _c(tasks)
    sc = scoring or default_scoring()
    res = np.zeros(len(tasks), INDEL_CIGAR_RESULT)
    cig = np.zeros((len(tasks), cigar_stride), np.uint8)
    rc = self.lib.ref_indel_ext_cigar_batch(_p(genome), _p(reads), _p(offsets), _p(lens),
                                            C.c_int(encoding), _p(tasks), C.c_uint32(len(tasks)),
                                            C.c_uint32(max_indel), C.byref(sc), _p(res), _p(cig),
                                            C.c_uint32(cigar_stride))
    return rc, res, cig


def _ref_clip_boundary(self, records, cigars, scoring=None):
    sc = scoring or default_scoring()
    rec = np.array(records, CLIP_RECORD, copy=True)
    cig = np.array(cigars, np.uint8, copy=True)
    rc = self.lib.ref_clip_boundary_batch(C.c_uint32(len(rec)), _p(rec), _p(cig),
                                          C.c_uint32(cig.shape[1]), C.byref(sc))
    return rc, rec, cig


def _ref_serialize(self, mappings, direction):
    m = _c(mappings)
    d = _c(direction, np.uint8)
    out = np.zeros((len(m), 12), np.uint8)
    sizes = np.zeros(len(m), np.uint8)
    rc = self.lib.ref_serialize_batch(C.c_uint32(len(m)), _p(m), _p(d), _p(out), _p(sizes))
    return rc, out, sizes


Reference.indel_match = _ref_indel_match
Reference.indel_ext_cigar = _ref_indel_ext_cigar
Reference.clip_boundary = _ref_clip_boundary
Reference.serialize = _ref_serialize


def _ref_cf_filter(self, genome, ref_size, cur_offset, reads, offsets, lens, tasks, max_diff=20):
    """The reference's own calculate_CF_128b_* and test_CF_128b_SSE_difference."""
    tasks = np.ascontiguousarray(tasks, CF_TASK)
    res = np.zeros(len(tasks), np.uint32)
    ok = np.zeros(len(tasks), np.uint8)
    rc = self.lib.ref_cf_batch(_p(genome), C.c_uint32(ref_size), C.c_uint32(cur_offset), _p(reads), _p(offsets),
                               _p(lens), _p(tasks), C.c_uint32(len(tasks)), _p(res), _p(ok), C.c_uint32(max_diff))
    return rc, res, ok


Reference.cf_filter = _ref_cf_filter


def _ref_sam_cigar(self, ext_cigars, out_stride=None):
    return _sam_cigar(self.lib.ref_sam_cigar_batch, ext_cigars, out_stride)


Reference.sam_cigar = _ref_sam_cigar


# ---- row L: clipIllegalPositions / clipOverlappingPairs (soft_clipping.cpp:592-729, 997-1151)
CLIP_ILLEGAL_TASK = np.dtype([("ref_size", "<i4"), ("read_length", "<u2"), ("perfect", "<u2")])
CLIP_PAIR_TASK = np.dtype([("ref_seq_id", "<i4", (2,)), ("read_length", "<u2", (2,)), ("direct", "u1", (2,)),
                           ("perfect", "u1", (2,))])
assert CLIP_ILLEGAL_TASK.itemsize == 8 and CLIP_PAIR_TASK.itemsize == 16


def _clip_illegal(fn, tasks, records, cigars, scoring=None):
    sc = scoring or default_scoring()
    t = np.ascontiguousarray(tasks, CLIP_ILLEGAL_TASK)
    rec = np.array(records, CLIP_RECORD, copy=True)
    cig = np.array(cigars, np.uint8, copy=True)
    rc = fn(C.c_uint32(len(rec)), _p(t), _p(rec), _p(cig), C.c_uint32(cig.shape[1]), C.byref(sc))
    return rc, rec, cig


def _clip_pairs(fn, tasks, records, cigars, *extra):
    t = np.ascontiguousarray(tasks, CLIP_PAIR_TASK)
    rec = np.array(records, CLIP_RECORD, copy=True)
    cig = np.array(cigars, np.uint8, copy=True)
    assert len(rec) == 2 * len(t) and len(cig) == len(rec)
    rc = fn(C.c_uint32(len(t)), _p(t), _p(rec), _p(cig), C.c_uint32(cig.shape[1]), *extra)
    return rc, rec, cig


def _o_clip_illegal(self, tasks, records, cigars, scoring=None):
    return _clip_illegal(self.lib.wo_clip_illegal_batch, tasks, records, cigars, scoring)


def _o_clip_pairs(self, tasks, records, cigars):
    return _clip_pairs(self.lib.wo_clip_pairs_batch, tasks, records, cigars)


def _r_clip_illegal(self, tasks, records, cigars, scoring=None):
    return _clip_illegal(self.lib.ref_clip_illegal_batch, tasks, records, cigars, scoring)


def _r_clip_pairs(self, tasks, records, cigars, scoring=None):
    sc = scoring or default_scoring()
    return _clip_pairs(self.lib.ref_clip_pairs_batch, tasks, records, cigars, C.byref(sc))


Oracle.clip_illegal = _o_clip_illegal
Oracle.clip_overlapping_pairs = _o_clip_pairs
Reference.clip_illegal = _r_clip_illegal
Reference.clip_overlapping_pairs = _r_clip_pairs


# ---- row S: CSoftClipping::match + getExtCigar (soft_clipping.cpp:139-589)
SOFTCLIP_TASK = np.dtype([("read_idx", "<u4"), ("ref_pos", "<u4"), ("max_distance_in_ref", "<u4"),
                          ("min_exact_len", "<u2"), ("flags", "u1"), ("reserved", "u1")])
SOFTCLIP_RESULT = np.dtype([("affine_score", "<f8"), ("found", "<u4"), ("pos", "<u4"), ("left_clipping", "<u4"),
                            ("right_clipping", "<u4"), ("left_match", "<u4"), ("right_match", "<u4"),
                            ("del_size", "<i4"), ("no_mismatches", "<u4"), ("cigar_len", "<u4"), ("reserved", "<u4")])
assert SOFTCLIP_TASK.itemsize == 16 and SOFTCLIP_RESULT.itemsize == 48


class SoftclipParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("max_query_len", "min_approx_indel_len", "max_approx_indel_len",
                                          "max_approx_indel_mismatches", "fresh_object", "reserved")]


def default_softclip_params(max_query_len=151, max_mismatches=12, fresh_object=0):
    # mapper/params.h:134-136, 174: 3, 50, max_no_errors * sensitivity_factor
    return SoftclipParams(max_query_len, 3, 50, max_mismatches, fresh_object, 0)


def _o_softclip(self, genome, reads, offsets, lens, encoding, tasks, params=None, scoring=None, cigar_stride=512):
    tasks = np.ascontiguousarray(tasks, SOFTCLIP_TASK)
    prm = params or default_softclip_params()
    sc = scoring or default_scoring()
    res = np.zeros(len(tasks), SOFTCLIP_RESULT)
    cig = np.zeros((len(tasks), cigar_stride), np.uint8)
    rc = self.lib.wo_softclip_batch(_p(genome), _p(reads), _p(offsets), _p(lens), C.c_int(encoding), _p(tasks),
                                    C.c_uint32(len(tasks)), C.byref(prm), C.byref(sc), _p(res), _p(cig),
                                    C.c_uint32(cigar_stride))
    return rc, res, cig


def _r_softclip(self, genome, reads, offsets, lens, encoding, tasks, params=None, scoring=None, cigar_stride=512,
                max_text_len=16384):
    tasks = np.ascontiguousarray(tasks, SOFTCLIP_TASK)
    prm = params or default_softclip_params()
    sc = scoring or default_scoring()
    res = np.zeros(len(tasks), SOFTCLIP_RESULT)
    cig = np.zeros((len(tasks), cigar_stride), np.uint8)
    rc = self.lib.ref_softclip_batch(_p(genome), _p(reads), _p(offsets), _p(lens), C.c_int(encoding), _p(tasks),
                                     C.c_uint32(len(tasks)), C.byref(prm), C.byref(sc), C.c_uint32(max_text_len),
                                     _p(res), _p(cig), C.c_uint32(cigar_stride))
    return rc, res, cig


Oracle.softclip_match = _o_softclip
Reference.softclip_match = _r_softclip


# ---- the chosen pair: fill_cigar + clipping passes + SAM CIGAR / MD:Z (sam.cpp:1871-1900, 1992-2007)
def _finish_out(n_slots, ext_stride, out_stride):
    return (np.zeros(n_slots, FINISH_RESULT), np.zeros((n_slots, ext_stride), np.uint8),
            np.zeros((n_slots, out_stride), np.uint8), np.zeros((n_slots, out_stride), np.uint8))


def _o_finish_pairs(self, genome, reads, offsets, lens, tasks, params=None, scoring=None, ext_stride=512):
    t = np.ascontiguousarray(tasks, FINISH_TASK)
    prm = params or default_finish_params()
    sc = scoring or default_scoring()
    res, ext, cig, mdz = _finish_out(len(t), ext_stride, 2 * ext_stride + 16)
    rc = self.lib.wo_finish_pairs_batch(_p(genome), _p(reads), _p(offsets), _p(lens), _p(t), C.c_uint32(len(t) // 2),
                                        C.byref(prm), C.byref(sc), _p(res), _p(ext), C.c_uint32(ext.shape[1]), _p(cig),
                                        C.c_uint32(cig.shape[1]), _p(mdz), C.c_uint32(mdz.shape[1]))
    return rc, res, ext, cig, mdz


def _r_finish_pairs(self, genome, ref_size, desc, reads, offsets, lens, read_idx, wire, distant, stale_penalty,
                    params=None, scoring=None, max_approx_indel_len=50, ext_stride=512, model=(500.0, 125.0)):
    """The reference's own chain from the wire records on: returns (rc, tasks as the generator constructs them,
    results, ext-CIGARs, CIGARs, MD:Z)."""
    prm = params or default_finish_params()
    sc = scoring or default_scoring()
    n = len(read_idx)
    desc = np.ascontiguousarray(desc, np.uint64)
    tasks = np.zeros(n, FINISH_TASK)
    res, ext, cig, mdz = _finish_out(n, ext_stride, 2 * ext_stride + 16)
    rc = self.lib.ref_finish_pairs_batch(_p(genome), C.c_uint32(ref_size), _p(desc), C.c_uint32(len(desc)), _p(reads),
                                         _p(offsets), _p(lens), C.c_uint32(n // 2), _p(_c(read_idx, np.uint32)),
                                         _p(_c(wire, np.uint8)), _p(_c(distant, np.uint8)),
                                         _p(_c(stale_penalty, np.uint32)), C.byref(prm), C.byref(sc),
                                         C.c_uint32(max_approx_indel_len), C.c_double(model[0]), C.c_double(model[1]),
                                         _p(tasks), _p(res), _p(ext),
                                         C.c_uint32(ext.shape[1]), _p(cig), C.c_uint32(cig.shape[1]), _p(mdz),
                                         C.c_uint32(mdz.shape[1]))
    return rc, tasks, res, ext, cig, mdz


Oracle.finish_pairs = _o_finish_pairs
Reference.finish_pairs = _r_finish_pairs


def _r_preprocess_only(self, reads, offsets, lens, encoding, max_query_len=255, max_text_len=180, nthreads=1):
    """LevMyers::preprocess once per read and nothing else (timing aid of bench.py's cpu_baseline)."""
    return self.lib.ref_preprocess_batch(_p(reads), _p(offsets), _p(lens), C.c_uint32(len(lens)), C.c_int(encoding),
                                         C.c_uint32(max_query_len), C.c_uint32(max_text_len), C.c_int(nthreads))


Reference.preprocess_only = _r_preprocess_only


# ---- verified candidates -> result-group bytes (mapping_core_sa.cpp:241-344, mapping_core.cpp:976-997, results.cpp)
def _commit(fn, tasks, indel):
    t = np.ascontiguousarray(tasks, COMMIT_TASK)
    out = np.zeros(len(t), MAPPING)
    ind = None if indel is None else np.ascontiguousarray(indel, MAPPING)
    rc = fn(C.c_uint32(len(t)), _p(t), _p(ind) if ind is not None else C.c_void_p(0), _p(out))
    return rc, out


def _groups(fn, tasks, committed, read_first, read_ids, params, capacity=None):
    t = np.ascontiguousarray(tasks, COMMIT_TASK)
    m = np.ascontiguousarray(committed, MAPPING)
    rf = np.ascontiguousarray(read_first, np.uint32)
    ids = np.ascontiguousarray(read_ids, np.uint64)
    cap = capacity or (len(ids) * 16 + len(t) * 11 + 64)
    out = np.zeros(cap, np.uint8)
    nb = C.c_uint32(0)
    rc = fn(C.c_uint32(len(t)), _p(t), _p(m), C.c_uint32(len(ids)), _p(rf), _p(ids), C.byref(params), _p(out),
            C.c_uint32(cap), C.byref(nb))
    return rc, out[:nb.value].copy()


Oracle.commit = lambda self, tasks, indel=None: _commit(self.lib.wo_commit_batch, tasks, indel)
Reference.commit = lambda self, tasks, indel=None: _commit(self.lib.ref_commit_batch, tasks, indel)
Oracle.result_groups = lambda self, *a, **k: _groups(self.lib.wo_result_groups_batch, *a, **k)
Reference.result_groups = lambda self, *a, **k: _groups(self.lib.ref_result_groups_batch, *a, **k)


# ---- the suffix-array loops on the reference's own CMappingCore (mapping_core_sa.cpp:193-513)
def _r_search_reads(self, genome, ref_size, reads, offsets, lens, sa_dir, sa_rc, seg, min_read_len, k, sensitive=0,
                    enable_indels=0, max_indel=50, params=None, sa_skip=64, capacity=None):
    """Returns (rc, ranges [n, 4], best [n, 2], pushed mappings, their strand, their read)."""
    prm = params or default_indel_params()
    n = len(lens)
    sa_dir = np.ascontiguousarray(sa_dir, np.uint32)
    sa_rc = np.ascontiguousarray(sa_rc, np.uint32)
    cap = capacity or (64 * n + 1024)
    ranges = np.zeros((n, 4), np.uint32)
    best = np.zeros((n, 2), np.uint32)
    pushed = np.zeros(cap, MAPPING)
    pdir = np.zeros(cap, np.uint8)
    pread = np.zeros(cap, np.uint32)
    cnt = C.c_uint32(0)
    rc = self.lib.ref_search_reads_batch(_p(genome), C.c_uint32(ref_size), _p(reads), _p(offsets), _p(lens), C.c_uint32(n),
                                         _p(sa_dir), C.c_uint32(len(sa_dir)), _p(sa_rc), C.c_uint32(len(sa_rc)),
                                         C.c_uint32(seg[0]), C.c_uint32(seg[1]), C.c_uint32(min_read_len), C.c_uint32(k),
                                         C.c_int(sensitive), C.c_int(enable_indels), C.c_uint32(max_indel), C.byref(prm),
                                         C.c_uint32(sa_skip), _p(ranges), _p(best), _p(pushed), _p(pdir), _p(pread),
                                         C.c_uint32(cap), C.byref(cnt))
    return rc, ranges, best, pushed[:cnt.value].copy(), pdir[:cnt.value].copy(), pread[:cnt.value].copy()


Reference.search_reads = _r_search_reads


# ---- mate rescue by LevMyers: find_pairs<LevMyers>'s chain per hit (sam_matching.cpp:379-457)
RESCUE_TASK = np.dtype([("mate_read_idx", "<u4"), ("hit_pos", "<u4"), ("distance_threshold", "<u4"), ("seq_lo", "<u4"),
                        ("seq_hi", "<u4"), ("seq_delta", "<i4"), ("ref_size", "<i4"), ("max_mate_edit_distance", "<u2"),
                        ("hit_num_events", "<u2"), ("paired_dir_direct", "u1"), ("reserved", "u1", (3,))])
RESCUE_RESULT = np.dtype([("score", "<f8"), ("ok", "<u4"), ("aligned", "<u2"), ("edit_distance", "<u2"), ("pos", "<u4"), ("ref_seq_pos", "<i4"),
                          ("ref_length", "<u2"), ("err_edit_distance", "<u2"), ("left_clipping", "<u2"),
                          ("right_clipping", "<u2"), ("ext_cigar_len", "<u4"), ("cigar_events", "<u4")])
assert RESCUE_TASK.itemsize == 36 and RESCUE_RESULT.itemsize == 40


class RescueParams(C.Structure):
    _fields_ = [("enable_short_indel_refinement", C.c_uint32), ("enable_boundary_clipping", C.c_uint32)]


def _o_rescue_mates(self, genome, reads, offsets, lens, tasks, params=None, scoring=None, ext_stride=512):
    prm = params or RescueParams(1, 1)
    sc = scoring or default_scoring()
    t = np.ascontiguousarray(tasks, RESCUE_TASK)
    res = np.zeros(len(t), RESCUE_RESULT)
    ext = np.zeros((len(t), ext_stride), np.uint8)
    rc = self.lib.wo_rescue_mates_batch(_p(genome), _p(reads), _p(offsets), _p(lens), _p(t), C.c_uint32(len(t)), C.byref(prm),
                                        C.byref(sc), _p(res), _p(ext), C.c_uint32(ext_stride))
    return rc, res, ext


def _r_rescue_mates(self, genome, ref_size, desc, reads, offsets, lens, mate_read_idx, hit_pos, hit_seq_id, hit_num_events,
                    paired_dir_direct, max_mate_ed, params=None, scoring=None, model=(500.0, 125.0), ext_stride=512):
    """The reference's own statements on its own objects: returns (rc, tasks as the C-ABI takes them, results, ext-CIGARs)."""
    prm = params or RescueParams(1, 1)
    sc = scoring or default_scoring()
    n = len(mate_read_idx)
    desc = np.ascontiguousarray(desc, np.uint64)
    tasks = np.zeros(n, RESCUE_TASK)
    res = np.zeros(n, RESCUE_RESULT)
    ext = np.zeros((n, ext_stride), np.uint8)
    rc = self.lib.ref_rescue_mates_batch(_p(genome), C.c_uint32(ref_size), _p(desc), C.c_uint32(len(desc)), _p(reads),
                                         _p(offsets), _p(lens), C.c_uint32(n), _p(_c(mate_read_idx, np.uint32)),
                                         _p(_c(hit_pos, np.uint32)), _p(_c(hit_seq_id, np.int32)),
                                         _p(_c(hit_num_events, np.uint16)), _p(_c(paired_dir_direct, np.uint8)),
                                         _p(_c(max_mate_ed, np.uint16)), C.byref(prm), C.byref(sc), C.c_double(model[0]),
                                         C.c_double(model[1]), _p(tasks), _p(res), _p(ext), C.c_uint32(ext_stride))
    return rc, tasks, res, ext


Oracle.rescue_mates = _o_rescue_mates
Reference.rescue_mates = _r_rescue_mates
