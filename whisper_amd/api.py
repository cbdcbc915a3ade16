"""Python binding of the C-ABI (include/whisper_amd.h) - the host-side mirror used by
the parity tests and bench.py.  It loads whisper_amd/libwhisper_amd.so and nothing
else: there is no CPU fallback, a missing library or GPU raises.

Class names follow the reference objects whose calls are batched:
  Context   <- setReference() of every verifier (LevMyers64.cpp:83-89, edit_dist.cpp:98-103)
  Reads     <- preprocess() / preprocessRawSeq() (LevMyers64.cpp:167-220)
  Context.dynamic_programming <- LevMyers::dynamicProgramming (LevMyers.h:51)
  Context.lev_diag            <- CEditDist::LevDiag (edit_dist.h:50-51)
  Context.align               <- dynamicProgramming + getExtCigar + refine_ext_cigar
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwhisper_amd.so")

VARIANT_AUTO, VARIANT_M64, VARIANT_M128, VARIANT_M256 = 0, 1, 2, 3
READS_NIBBLE, READS_ASCII = 0, 1
FLAG_REV_COMP = 1
LAST_SYMBOL_FROM_READ = -1
DISTANCE_EXACT, DISTANCE_ACCEPTED_ONLY = 0, 1

DIST_TASK = np.dtype([("read_idx", "<u4"), ("ref_pos", "<u4"), ("max_distance_in_ref", "<u4"),
                      ("max_ed", "<u2"), ("variant", "u1"), ("flags", "u1")])
DIST_RESULT = np.dtype([("edit_distance", "<u4"), ("pos", "<u4")])
LEVDIAG_TASK = np.dtype([("read_idx", "<u4"), ("position_in_text", "<u4"), ("fixed_left", "<u2"),
                         ("fixed_right", "<u2"), ("max_mismatches", "<u2"),
                         ("dir_gen_flag", "<u2")])
ALIGN_RESULT = np.dtype([("affine_score", "<f8"), ("ok", "<u4"), ("edit_distance", "<u4"),
                         ("end_pos", "<u4"), ("begin_pos", "<u4"), ("num_events", "<u4"),
                         ("cigar_len", "<u4"), ("refined", "<u4"), ("reserved", "<u4")])

CF_TASK = np.dtype([("read_idx", "<u4"), ("position", "<u4"), ("read_offset", "<u2"), ("pattern_len", "<u2"),
                    ("fixed_left", "<u2"), ("fixed_right", "<u2"), ("dir_gen_flag", "u1"), ("flags", "u1"),
                    ("reserved", "<u2")])
CLIP_ILLEGAL_TASK = np.dtype([("ref_size", "<i4"), ("read_length", "<u2"), ("perfect", "<u2")])
CLIP_PAIR_TASK = np.dtype([("ref_seq_id", "<i4", (2,)), ("read_length", "<u2", (2,)), ("direct", "u1", (2,)),
                           ("perfect", "u1", (2,))])
SOFTCLIP_TASK = np.dtype([("read_idx", "<u4"), ("ref_pos", "<u4"), ("max_distance_in_ref", "<u4"),
                          ("min_exact_len", "<u2"), ("flags", "u1"), ("reserved", "u1")])
SOFTCLIP_RESULT = np.dtype([("affine_score", "<f8"), ("found", "<u4"), ("pos", "<u4"), ("left_clipping", "<u4"),
                            ("right_clipping", "<u4"), ("left_match", "<u4"), ("right_match", "<u4"),
                            ("del_size", "<i4"), ("no_mismatches", "<u4"), ("cigar_len", "<u4"), ("reserved", "<u4")])


class SoftclipParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("max_query_len", "min_approx_indel_len", "max_approx_indel_len",
                                          "max_approx_indel_mismatches", "fresh_object", "reserved")]


def default_softclip_params(max_query_len=151, max_mismatches=12, fresh_object=0):
    # mapper/params.h:134-136, 174
    return SoftclipParams(max_query_len, 3, 50, max_mismatches, fresh_object, 0)


SAM_CIGAR_LEN = np.dtype([("cigar_len", "<u4"), ("mdz_len", "<u4"), ("cigar_bin_len", "<u4"), ("reserved", "<u4")])
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


RESCUE_TASK = np.dtype([("mate_read_idx", "<u4"), ("hit_pos", "<u4"), ("distance_threshold", "<u4"), ("seq_lo", "<u4"),
                        ("seq_hi", "<u4"), ("seq_delta", "<i4"), ("ref_size", "<i4"), ("max_mate_edit_distance", "<u2"),
                        ("hit_num_events", "<u2"), ("paired_dir_direct", "u1"), ("reserved", "u1", (3,))])
RESCUE_RESULT = np.dtype([("score", "<f8"), ("ok", "<u4"), ("aligned", "<u2"), ("edit_distance", "<u2"), ("pos", "<u4"), ("ref_seq_pos", "<i4"),
                          ("ref_length", "<u2"), ("err_edit_distance", "<u2"), ("left_clipping", "<u2"),
                          ("right_clipping", "<u2"), ("ext_cigar_len", "<u4"), ("cigar_events", "<u4")])
assert RESCUE_TASK.itemsize == 36 and RESCUE_RESULT.itemsize == 40


class RescueParams(C.Structure):
    _fields_ = [("enable_short_indel_refinement", C.c_uint32), ("enable_boundary_clipping", C.c_uint32)]


class SaBatch(C.Structure):
    """wamd_sa_batch: what a stage of the suffix-array loops holds fixed (mapping_core_sa.cpp:193-348)."""
    _fields_ = [("read_first", C.c_uint32), ("n_reads", C.c_uint32), ("reverse_read_offset", C.c_uint32),
                ("max_mismatches", C.c_uint16), ("fixed_left", C.c_uint16), ("fixed_right", C.c_uint16),
                ("min_read_len", C.c_uint16), ("verifier", C.c_uint8), ("reserved", C.c_uint8 * 3)]


SA_VERIFIER_DISTANCE, SA_VERIFIER_LEVDIAG = 0, 1


def sa_batch(n_reads, k, seg, min_read_len, verifier, read_first=0, reverse_read_offset=0):
    b = SaBatch()
    b.read_first, b.n_reads, b.reverse_read_offset = read_first, n_reads, reverse_read_offset
    b.max_mismatches, b.fixed_left, b.fixed_right, b.min_read_len, b.verifier = k, seg[0], seg[1], min_read_len, verifier
    return b


class PinnedBuffer:
    """Page-locked host memory from wamd_host_alloc, seen as a numpy array."""

    def __init__(self, n, dtype):
        self.p = C.c_void_p()
        dt = np.dtype(dtype)
        _check(lib().wamd_host_alloc(C.byref(self.p), C.c_uint64(max(1, n) * dt.itemsize)))
        self.array = np.ctypeslib.as_array(C.cast(self.p, C.POINTER(C.c_uint8)), shape=(max(1, n) * dt.itemsize,)).view(dt)[:n]

    def free(self):
        if self.p:
            self.array = None
            lib().wamd_host_free(self.p)
            self.p = None


class GroupParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("id_bytes", "mapping_counter_size", "max_no_mappings", "reserved")]


class FinishParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("flank", "enable_short_indel_refinement", "enable_boundary_clipping",
                                          "enable_paired_clipping")]


def default_finish_params(flank=6):
    # mapper/params.h: refinement, boundary and paired clipping on; flank = max_no_errors (+ sensitivity) + 1
    return FinishParams(flank, 1, 1, 1)


STATUS = {0: "WAMD_OK", 1: "WAMD_ERR_INVALID_ARG", 2: "WAMD_ERR_HIP", 3: "WAMD_ERR_NO_DEVICE",
          4: "WAMD_ERR_TOO_LONG", 5: "WAMD_ERR_TEXT_LEN", 6: "WAMD_ERR_RANGE",
          7: "WAMD_ERR_NO_MEMORY", 8: "WAMD_ERR_CAPACITY"}

EXPORTS = [
    "wamd_version", "wamd_last_error", "wamd_device_count", "wamd_ctx_create",
    "wamd_sa_candidates_workspace_bytes", "wamd_verify_sa_candidates_dev", "wamd_verify_sa_candidates", "wamd_host_alloc",
    "wamd_host_free", "wamd_reads_upload_stream", "wamd_reads_refill", "wamd_reads_refill_async", "wamd_rescue_workspace_bytes", "wamd_rescue_mates",
    "wamd_rescue_mates_dev",
    "wamd_ctx_create_dev", "wamd_ctx_destroy", "wamd_ctx_set_scoring",
    "wamd_ctx_set_max_text_len", "wamd_ctx_set_distance_mode", "wamd_ctx_set_reference_params", "wamd_ctx_stream",
    "wamd_ctx_synchronize", "wamd_ctx_take_flags", "wamd_reads_upload", "wamd_reads_free",
    "wamd_reads_count", "wamd_verify_distance", "wamd_verify_distance_dev", "wamd_levdiag",
    "wamd_levdiag_dev", "wamd_cf_filter", "wamd_cf_filter_dev", "wamd_align", "wamd_align_dev", "wamd_align_workspace_bytes", "wamd_align_workspace_bytes_for",
    "wamd_indel_match", "wamd_indel_match_dev", "wamd_indel_ext_cigar", "wamd_indel_ext_cigar_dev",
    "wamd_clip_boundary", "wamd_clip_boundary_dev", "wamd_clip_illegal", "wamd_clip_illegal_dev",
    "wamd_clip_overlapping_pairs", "wamd_clip_overlapping_pairs_dev", "wamd_softclip_match",
    "wamd_softclip_match_dev", "wamd_serialize_mappings",
    "wamd_serialize_mappings_dev", "wamd_sam_cigar", "wamd_sam_cigar_dev", "wamd_records_scratch_bytes", "wamd_emit_lev_records_dev",
    "wamd_emit_lev_records_static_dev", "wamd_finish_workspace_bytes", "wamd_finish_pairs", "wamd_finish_pairs_dev",
    "wamd_comm_get_unique_id", "wamd_comm_create", "wamd_comm_create_all", "wamd_comm_destroy",
    "wamd_gather_records_dev", "wamd_gather_records_all", "wamd_commit_candidates_dev",
    "wamd_result_groups_workspace_bytes", "wamd_emit_result_groups_dev",
    "wamd_filter_verify_workspace_bytes", "wamd_filter_verify_distance_dev", "wamd_filter_levdiag_dev",
]


class Scoring(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("match", "mismatch", "gap_ins_open", "gap_ins_extend",
                                           "gap_del_open", "gap_del_extend", "clipping")]


class IndelParams(C.Structure):
    _fields_ = [("scoring", Scoring), ("min_clipped_factor", C.c_double), ("max_indel_bound", C.c_uint32),
                ("reserved", C.c_uint32)]


def default_scoring():
    # mapper/params.h:145-154
    return Scoring(1.0, -5.0, -5.0, -0.4, -5.0, -0.4, -6.0)


def default_indel_params(min_clipped_factor=1.0, scoring=None):
    # mapper/params.h:145-158
    return IndelParams(scoring or default_scoring(), min_clipped_factor, 0, 0)


class WamdError(RuntimeError):
    def __init__(self, status, text):
        super().__init__("%s: %s" % (STATUS.get(status, status), text))
        self.status = status


_lib = None


def lib():
    """The loaded C-ABI library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                LIB_PATH + " is missing: build it with `make -C whisper_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.wamd_version.restype = C.c_char_p
        L.wamd_last_error.restype = C.c_char_p
        L.wamd_ctx_stream.restype = C.c_void_p
        L.wamd_align_workspace_bytes.restype = C.c_uint64
        L.wamd_align_workspace_bytes_for.restype = C.c_uint64
        L.wamd_reads_count.restype = C.c_uint32
        L.wamd_records_scratch_bytes.restype = C.c_uint64
        L.wamd_finish_workspace_bytes.restype = C.c_uint64
        L.wamd_result_groups_workspace_bytes.restype = C.c_uint64
        L.wamd_filter_verify_workspace_bytes.restype = C.c_uint64
        L.wamd_sa_candidates_workspace_bytes.restype = C.c_uint64
        L.wamd_rescue_workspace_bytes.restype = C.c_uint64
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise WamdError(rc, lib().wamd_last_error().decode())


def _ptr(a):
    """void* of a numpy array (host) or of a torch tensor (host or device)."""
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(a.data_ptr())


class Reads:
    def __init__(self, ctx, data, offsets, lens, encoding, orientations=3, stream=None):
        self.ctx = ctx
        data = np.ascontiguousarray(data, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        lens = np.ascontiguousarray(lens, np.uint32)
        self.n = len(lens)
        self.encoding = encoding
        self.max_len = int(lens.max()) if self.n else 0
        self.h = C.c_void_p()
        if stream is None:
            _check(lib().wamd_reads_upload(ctx.h, _ptr(data), _ptr(offsets), _ptr(lens),
                                           C.c_uint32(self.n), C.c_int(encoding),
                                           C.c_int(orientations), C.byref(self.h)))
        else:   # beside another bin's kernels: waits for `stream` only
            _check(lib().wamd_reads_upload_stream(ctx.h, _ptr(data), _ptr(offsets), _ptr(lens),
                                                  C.c_uint32(self.n), C.c_int(encoding),
                                                  C.c_int(orientations), C.c_void_p(stream), C.byref(self.h)))

    def refill(self, data, offsets, lens, stream=None):
        """The next bin's reads into this batch's buffers (wamd_reads_refill)."""
        data = np.ascontiguousarray(data, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        lens = np.ascontiguousarray(lens, np.uint32)
        self.n = len(lens)
        self.max_len = int(lens.max()) if self.n else 0
        _check(lib().wamd_reads_refill(self.h, _ptr(data), _ptr(offsets), _ptr(lens), C.c_uint32(self.n),
                                       C.c_void_p(stream or 0)))

    def refill_async(self, data, data_bytes, offsets, lens, max_len, stream=None):
        """wamd_reads_refill_async: no host pass over the reads, no wait.  The arrays are the caller's (page-locked ones
        make the copies asynchronous) and must stay untouched until `stream` has passed the call; they are kept referenced
        here until the next refill."""
        assert data.dtype == np.uint8 and offsets.dtype == np.uint64 and lens.dtype == np.uint32
        assert data.flags.c_contiguous and offsets.flags.c_contiguous and lens.flags.c_contiguous
        self.n = len(lens)
        self.max_len = int(max_len)
        self._pending = (data, offsets, lens)
        _check(lib().wamd_reads_refill_async(self.h, _ptr(data), C.c_uint64(int(data_bytes)), _ptr(offsets), _ptr(lens),
                                             C.c_uint32(self.n), C.c_uint32(int(max_len)), C.c_void_p(stream or 0)))

    def free(self):
        if self.h:
            lib().wamd_reads_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    def __init__(self, genome, ref_size=None, cur_offset=0, device=0):
        """genome: packed reference bytes (numpy uint8 on the host, or a torch uint8 tensor
        already on the device, which is then borrowed, not copied)."""
        self.h = C.c_void_p()
        self._keep = genome
        if isinstance(genome, np.ndarray):
            genome = np.ascontiguousarray(genome, np.uint8)
            nbytes = genome.size
            fn = lib().wamd_ctx_create
        else:
            nbytes = genome.numel()
            fn = lib().wamd_ctx_create_dev
        if ref_size is None:
            ref_size = 2 * nbytes
        _check(fn(C.c_int(device), _ptr(genome), C.c_uint64(nbytes), C.c_uint32(ref_size),
                  C.c_uint32(cur_offset), C.byref(self.h)))

    def destroy(self):
        if self.h:
            lib().wamd_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def set_scoring(self, sc):
        _check(lib().wamd_ctx_set_scoring(self.h, C.byref(sc)))

    def set_test_hooks(self, hooks):
        """Not part of the C-ABI header: 1 = post passes without length hints, 2 = LevDiag by run extension only."""
        _check(lib().wamd_internal_set_test_hooks(self.h, C.c_uint32(hooks)))

    def set_max_text_len(self, v):
        _check(lib().wamd_ctx_set_max_text_len(self.h, C.c_uint32(v)))

    def set_distance_mode(self, mode):
        _check(lib().wamd_ctx_set_distance_mode(self.h, C.c_int(mode)))

    def set_reference_params(self, ref_size, cur_offset):
        _check(lib().wamd_ctx_set_reference_params(self.h, C.c_uint32(ref_size),
                                                   C.c_uint32(cur_offset)))

    def stream(self):
        return lib().wamd_ctx_stream(self.h)

    def synchronize(self):
        _check(lib().wamd_ctx_synchronize(self.h))

    def take_flags(self, stream=None):
        f = C.c_uint32(0)
        _check(lib().wamd_ctx_take_flags(self.h, C.c_void_p(stream or 0), C.byref(f)))
        return f.value

    def upload_reads(self, data, offsets, lens, encoding, orientations=3, stream=None):
        return Reads(self, data, offsets, lens, encoding, orientations, stream)

    # ---- host-pointer entry points ------------------------------------------------------
    def dynamic_programming(self, reads, tasks, last_symbol=LAST_SYMBOL_FROM_READ):
        tasks = np.ascontiguousarray(tasks, DIST_TASK)
        res = np.zeros(len(tasks), DIST_RESULT)
        _check(lib().wamd_verify_distance(self.h, reads.h, _ptr(tasks), C.c_uint32(len(tasks)),
                                          C.c_int(last_symbol), _ptr(res)))
        return res

    def lev_diag(self, reads, tasks):
        tasks = np.ascontiguousarray(tasks, LEVDIAG_TASK)
        res = np.zeros(len(tasks), np.uint32)
        _check(lib().wamd_levdiag(self.h, reads.h, _ptr(tasks), C.c_uint32(len(tasks)), _ptr(res)))
        return res

    def cf_filter(self, reads, tasks):
        """CF pre-filter distances for a batch (mapping_core_dp.cpp:349-483); pass iff <= 4k."""
        tasks = np.ascontiguousarray(tasks, CF_TASK)
        res = np.zeros(len(tasks), np.uint32)
        _check(lib().wamd_cf_filter(self.h, reads.h, _ptr(tasks), C.c_uint32(len(tasks)), _ptr(res)))
        return res

    def cf_filter_dev(self, reads, d_tasks, n, d_results, stream=None):
        _check(lib().wamd_cf_filter_dev(self.h, reads.h, _ptr(d_tasks), C.c_uint32(n), _ptr(d_results),
                                        C.c_void_p(stream or 0)))

    def filter_verify_workspace_bytes(self, n):
        return int(lib().wamd_filter_verify_workspace_bytes(C.c_uint32(n)))

    def filter_verify_distance_dev(self, reads, d_cf_tasks, d_tasks, n, d_results, d_ws, ws_bytes, last_symbol=-1,
                                   stream=None):
        """CF pre-filter, then dynamicProgramming on the candidates it passes (mapping_core_sa.cpp:223-255)."""
        _check(lib().wamd_filter_verify_distance_dev(self.h, reads.h, _ptr(d_cf_tasks), _ptr(d_tasks), C.c_uint32(n),
                                                     C.c_int(last_symbol), _ptr(d_results), _ptr(d_ws),
                                                     C.c_uint64(ws_bytes), C.c_void_p(stream or 0)))

    def filter_levdiag_dev(self, reads, d_cf_tasks, d_tasks, n, d_results, d_ws, ws_bytes, stream=None):
        """CF pre-filter, then LevDiag on the candidates it passes (mapping_core_sa.cpp:230-239)."""
        _check(lib().wamd_filter_levdiag_dev(self.h, reads.h, _ptr(d_cf_tasks), _ptr(d_tasks), C.c_uint32(n),
                                             _ptr(d_results), _ptr(d_ws), C.c_uint64(ws_bytes),
                                             C.c_void_p(stream or 0)))

    def sa_candidates_workspace_bytes(self, n_candidates, n_reads):
        return int(lib().wamd_sa_candidates_workspace_bytes(C.c_uint32(n_candidates), C.c_uint32(n_reads)))

    def verify_sa_candidates_dev(self, reads, batch, d_counts, d_positions, n, d_errors, d_end_positions, d_ws, ws_bytes,
                                 stream=None, d_read_k=None):
        """The suffix-array loops' candidates in their compact form (mapping_core_sa.cpp:193-348): counts per read and
        strand, 4-byte suffix-array entries; one byte of errors per candidate."""
        _check(lib().wamd_verify_sa_candidates_dev(self.h, reads.h, C.byref(batch), _ptr(d_counts), _ptr(d_read_k), _ptr(d_positions),
                                                   C.c_uint32(n), _ptr(d_errors),
                                                   _ptr(d_end_positions) if d_end_positions is not None else None,
                                                   _ptr(d_ws), C.c_uint64(ws_bytes), C.c_void_p(stream or 0)))

    def verify_sa_candidates(self, reads, batch, counts, positions, errors=None, end_positions=None, want_positions=False,
                             read_k=None):
        """Host buffers in, host buffers out: the chunked two-stream pipeline of wamd_verify_sa_candidates."""
        counts = np.ascontiguousarray(counts, np.uint16)
        positions = np.ascontiguousarray(positions, np.uint32)
        n = len(positions)
        if errors is None:
            errors = np.zeros(n, np.uint8)
        if end_positions is None and want_positions:
            end_positions = np.zeros(n, np.uint32)
        if read_k is not None:
            read_k = np.ascontiguousarray(read_k, np.uint8)
        _check(lib().wamd_verify_sa_candidates(self.h, reads.h, C.byref(batch), _ptr(counts), _ptr(read_k), _ptr(positions), C.c_uint32(n),
                                               _ptr(errors), _ptr(end_positions) if end_positions is not None else None))
        return (errors, end_positions) if end_positions is not None else errors

    def align(self, reads, tasks, last_symbol=0, refine=1, cigar_stride=1024):
        tasks = np.ascontiguousarray(tasks, DIST_TASK)
        res = np.zeros(len(tasks), ALIGN_RESULT)
        cig = np.zeros((len(tasks), cigar_stride), np.uint8)
        _check(lib().wamd_align(self.h, reads.h, _ptr(tasks), C.c_uint32(len(tasks)),
                                C.c_int(last_symbol), C.c_int(refine), _ptr(res), _ptr(cig),
                                C.c_uint32(cigar_stride)))
        return res, cig

    def indel_match(self, reads, tasks, params=None):
        """CIndelMatching::Match for a batch (indels.cpp:125-461)."""
        tasks = np.ascontiguousarray(tasks, INDEL_TASK)
        prm = params or default_indel_params()
        res = np.zeros(len(tasks), MAPPING)
        _check(lib().wamd_indel_match(self.h, reads.h, _ptr(tasks), C.c_uint32(len(tasks)),
                                      C.byref(prm), _ptr(res)))
        return res

    def indel_ext_cigar(self, reads, tasks, cigar_stride=1024):
        """CIndelMatching::GetExtCigarNew for a batch (indels.cpp:938-957)."""
        tasks = np.ascontiguousarray(tasks, INDEL_CIGAR_TASK)
        res = np.zeros(len(tasks), INDEL_CIGAR_RESULT)
        cig = np.zeros((len(tasks), cigar_stride), np.uint8)
        _check(lib().wamd_indel_ext_cigar(self.h, reads.h, _ptr(tasks), C.c_uint32(len(tasks)),
                                          _ptr(res), _ptr(cig), C.c_uint32(cigar_stride)))
        return res, cig

    def clip_boundary(self, records, cigars):
        """CSoftClipping::clipBoundaryPositions for a batch (soft_clipping.cpp:827-994)."""
        rec = np.array(records, CLIP_RECORD, copy=True)
        cig = np.array(cigars, np.uint8, copy=True)
        _check(lib().wamd_clip_boundary(self.h, C.c_uint32(len(rec)), _ptr(rec), _ptr(cig),
                                        C.c_uint32(cig.shape[1])))
        return rec, cig

    def clip_illegal(self, tasks, records, cigars):
        """CSoftClipping::clipIllegalPositions for a batch (soft_clipping.cpp:592-729)."""
        t = np.ascontiguousarray(tasks, CLIP_ILLEGAL_TASK)
        rec = np.array(records, CLIP_RECORD, copy=True)
        cig = np.array(cigars, np.uint8, copy=True)
        if len(t) != len(rec) or len(cig) != len(rec):
            raise ValueError("one task and one ext-CIGAR slot per record")
        _check(lib().wamd_clip_illegal(self.h, C.c_uint32(len(rec)), _ptr(t), _ptr(rec), _ptr(cig),
                                       C.c_uint32(cig.shape[1])))
        return rec, cig

    def clip_overlapping_pairs(self, tasks, records, cigars):
        """CSoftClipping::clipOverlappingPairs for a batch of pairs (soft_clipping.cpp:997-1151);
        pair i owns records and ext-CIGAR slots 2i (p.first) and 2i + 1 (p.second)."""
        t = np.ascontiguousarray(tasks, CLIP_PAIR_TASK)
        rec = np.array(records, CLIP_RECORD, copy=True)
        cig = np.array(cigars, np.uint8, copy=True)
        if len(rec) != 2 * len(t) or len(cig) != len(rec):
            raise ValueError("two records and two ext-CIGAR slots per pair")
        _check(lib().wamd_clip_overlapping_pairs(self.h, C.c_uint32(len(t)), _ptr(t), _ptr(rec), _ptr(cig),
                                                 C.c_uint32(cig.shape[1])))
        return rec, cig

    def softclip_match(self, reads, tasks, params=None, cigar_stride=512):
        """CSoftClipping::{preprocessRawSeq, match, getExtCigar} for a batch (soft_clipping.cpp:139-589)."""
        t = np.ascontiguousarray(tasks, SOFTCLIP_TASK)
        prm = params or default_softclip_params()
        res = np.zeros(len(t), SOFTCLIP_RESULT)
        cig = np.zeros((len(t), cigar_stride), np.uint8)
        _check(lib().wamd_softclip_match(self.h, reads.h, _ptr(t), C.c_uint32(len(t)), C.byref(prm), _ptr(res),
                                         _ptr(cig), C.c_uint32(cigar_stride)))
        return res, cig

    def softclip_match_dev(self, reads, d_tasks, n, d_results, d_cigars, cigar_stride, max_window, params=None,
                           stream=None):
        prm = params or default_softclip_params()
        _check(lib().wamd_softclip_match_dev(self.h, reads.h, _ptr(d_tasks), C.c_uint32(n), C.byref(prm),
                                             C.c_uint32(max_window), _ptr(d_results), _ptr(d_cigars),
                                             C.c_uint32(cigar_stride), C.c_void_p(stream or 0)))

    def sam_cigar(self, ext_cigars, out_stride=None):
        """extract_cigar_and_mdz / extract_cigarbin_and_mdz for a batch (sam.cpp:805-1051):
        returns CIGAR text, MD:Z, BAM operations and the lengths."""
        ext = np.ascontiguousarray(ext_cigars, np.uint8)
        n, es = ext.shape
        os_ = out_stride or 2 * es + 16
        cig = np.zeros((n, os_), np.uint8)
        mdz = np.zeros((n, os_), np.uint8)
        cbin = np.zeros((n, es + 4), np.uint32)
        lens = np.zeros(n, SAM_CIGAR_LEN)
        _check(lib().wamd_sam_cigar(self.h, C.c_uint32(n), _ptr(ext), C.c_uint32(es), _ptr(cig), C.c_uint32(os_),
                                    _ptr(mdz), C.c_uint32(os_), _ptr(cbin), C.c_uint32(es + 4), _ptr(lens)))
        return cig, mdz, cbin, lens

    def commit_candidates_dev(self, n, d_tasks, d_indel, d_committed, stream=None):
        _check(lib().wamd_commit_candidates_dev(self.h, C.c_uint32(n), _ptr(d_tasks), _ptr(d_indel), _ptr(d_committed),
                                                C.c_void_p(stream or 0)))

    def result_groups_workspace_bytes(self, n_tasks, n_reads):
        return int(lib().wamd_result_groups_workspace_bytes(C.c_uint32(n_tasks), C.c_uint32(n_reads)))

    def emit_result_groups_dev(self, n_tasks, d_tasks, d_committed, n_reads, d_read_first, d_read_ids, params, d_out,
                               out_capacity, d_out_bytes, d_read_off, d_ws, stream=None):
        _check(lib().wamd_emit_result_groups_dev(self.h, C.c_uint32(n_tasks), _ptr(d_tasks), _ptr(d_committed),
                                                 C.c_uint32(n_reads), _ptr(d_read_first), _ptr(d_read_ids),
                                                 C.byref(params), _ptr(d_out), C.c_uint32(out_capacity),
                                                 _ptr(d_out_bytes), _ptr(d_read_off), _ptr(d_ws),
                                                 C.c_void_p(stream or 0)))

    def finish_pairs(self, reads, tasks, params=None, ext_stride=512):
        """store_mapped_pair_reads' finish of the chosen pairs (sam.cpp:1871-1900, 1992-2007): returns results,
        ext-CIGARs, SAM CIGARs, MD:Z; slots 2i / 2i + 1 = p.first / p.second of pair i."""
        t = np.ascontiguousarray(tasks, FINISH_TASK)
        prm = params or default_finish_params()
        n = len(t)
        res = np.zeros(n, FINISH_RESULT)
        ext = np.zeros((n, ext_stride), np.uint8)
        cig = np.zeros((n, 2 * ext_stride + 16), np.uint8)
        mdz = np.zeros((n, 2 * ext_stride + 16), np.uint8)
        _check(lib().wamd_finish_pairs(self.h, reads.h, _ptr(t), C.c_uint32(n // 2), C.byref(prm), _ptr(res), _ptr(ext),
                                       C.c_uint32(ext_stride), _ptr(cig), C.c_uint32(cig.shape[1]), _ptr(mdz),
                                       C.c_uint32(mdz.shape[1])))
        return res, ext, cig, mdz

    def rescue_mates(self, reads, tasks, params=None, ext_stride=512):
        """find_pairs<LevMyers>'s chain per hit (sam_matching.cpp:379-457): results and the mates' ext-CIGARs."""
        t = np.ascontiguousarray(tasks, RESCUE_TASK)
        prm = params or RescueParams(1, 1)
        res = np.zeros(len(t), RESCUE_RESULT)
        ext = np.zeros((len(t), ext_stride), np.uint8)
        _check(lib().wamd_rescue_mates(self.h, reads.h, _ptr(t), C.c_uint32(len(t)), C.byref(prm), _ptr(res), _ptr(ext),
                                       C.c_uint32(ext_stride)))
        return res, ext

    def rescue_workspace_bytes(self, reads, n, max_ed):
        return int(lib().wamd_rescue_workspace_bytes(reads.h, C.c_uint32(n), C.c_uint32(max_ed)))

    def rescue_mates_dev(self, reads, d_tasks, n, d_results, d_ext, ext_stride, d_ws, ws_bytes, params=None, stream=None):
        prm = params or RescueParams(1, 1)
        _check(lib().wamd_rescue_mates_dev(self.h, reads.h, _ptr(d_tasks), C.c_uint32(n), C.byref(prm), _ptr(d_results),
                                           _ptr(d_ext), C.c_uint32(ext_stride), _ptr(d_ws), C.c_uint64(ws_bytes),
                                           C.c_void_p(stream or 0)))

    def finish_workspace_bytes(self, reads, n_pairs):
        return int(lib().wamd_finish_workspace_bytes(reads.h, C.c_uint32(n_pairs)))

    def finish_pairs_dev(self, reads, d_tasks, n_pairs, d_results, d_ext, ext_stride, d_cigars, cigar_stride, d_mdz,
                         mdz_stride, d_ws, ws_bytes, params=None, stream=None):
        prm = params or default_finish_params()
        _check(lib().wamd_finish_pairs_dev(self.h, reads.h, _ptr(d_tasks), C.c_uint32(n_pairs), C.byref(prm),
                                           _ptr(d_results), _ptr(d_ext), C.c_uint32(ext_stride), _ptr(d_cigars),
                                           C.c_uint32(cigar_stride), _ptr(d_mdz), C.c_uint32(mdz_stride), _ptr(d_ws),
                                           C.c_uint64(ws_bytes), C.c_void_p(stream or 0)))

    def serialize_mappings(self, mappings, direction):
        """candidate_mapping_t::serialize for a batch (results.cpp:161-217)."""
        m = np.ascontiguousarray(mappings, MAPPING)
        d = np.ascontiguousarray(direction, np.uint8)
        out = np.zeros((len(m), 12), np.uint8)
        sizes = np.zeros(len(m), np.uint8)
        _check(lib().wamd_serialize_mappings(self.h, C.c_uint32(len(m)), _ptr(m), _ptr(d), _ptr(out),
                                             _ptr(sizes)))
        return out, sizes

    # ---- device-pointer entry points (torch tensors on this device) ----------------------
    def dynamic_programming_dev(self, reads, d_tasks, n, d_results, last_symbol=-1, stream=None):
        _check(lib().wamd_verify_distance_dev(self.h, reads.h, _ptr(d_tasks), C.c_uint32(n),
                                              C.c_int(last_symbol), _ptr(d_results),
                                              C.c_void_p(stream or 0)))

    def lev_diag_dev(self, reads, d_tasks, n, d_results, stream=None):
        _check(lib().wamd_levdiag_dev(self.h, reads.h, _ptr(d_tasks), C.c_uint32(n),
                                      _ptr(d_results), C.c_void_p(stream or 0)))

    def indel_ext_cigar_dev(self, reads, d_tasks, n, d_results, d_cigars, cigar_stride, stream=None):
        _check(lib().wamd_indel_ext_cigar_dev(self.h, reads.h, _ptr(d_tasks), C.c_uint32(n), _ptr(d_results),
                                              _ptr(d_cigars), C.c_uint32(cigar_stride), C.c_void_p(stream or 0)))

    def clip_boundary_dev(self, n, d_records, d_cigars, cigar_stride, stream=None):
        _check(lib().wamd_clip_boundary_dev(self.h, C.c_uint32(n), _ptr(d_records), _ptr(d_cigars),
                                            C.c_uint32(cigar_stride), C.c_void_p(stream or 0)))

    def clip_illegal_dev(self, n, d_tasks, d_records, d_cigars, cigar_stride, stream=None):
        _check(lib().wamd_clip_illegal_dev(self.h, C.c_uint32(n), _ptr(d_tasks), _ptr(d_records), _ptr(d_cigars),
                                           C.c_uint32(cigar_stride), C.c_void_p(stream or 0)))

    def clip_overlapping_pairs_dev(self, n_pairs, d_tasks, d_records, d_cigars, cigar_stride, stream=None):
        _check(lib().wamd_clip_overlapping_pairs_dev(self.h, C.c_uint32(n_pairs), _ptr(d_tasks), _ptr(d_records),
                                                     _ptr(d_cigars), C.c_uint32(cigar_stride), C.c_void_p(stream or 0)))

    def sam_cigar_dev(self, n, d_ext, ext_stride, d_cigars, cigar_stride, d_mdz, mdz_stride, d_lens, d_bin=None,
                      bin_stride=0, stream=None):
        _check(lib().wamd_sam_cigar_dev(self.h, C.c_uint32(n), _ptr(d_ext), C.c_uint32(ext_stride), _ptr(d_cigars),
                                        C.c_uint32(cigar_stride), _ptr(d_mdz), C.c_uint32(mdz_stride), _ptr(d_bin),
                                        C.c_uint32(bin_stride), _ptr(d_lens), C.c_void_p(stream or 0)))

    def serialize_mappings_dev(self, n, d_mappings, d_direction, d_out, d_sizes, stream=None):
        _check(lib().wamd_serialize_mappings_dev(self.h, C.c_uint32(n), _ptr(d_mappings), _ptr(d_direction), _ptr(d_out),
                                                 _ptr(d_sizes), C.c_void_p(stream or 0)))

    def records_scratch_bytes(self, n):
        return int(lib().wamd_records_scratch_bytes(C.c_uint32(n)))

    def emit_lev_records_dev(self, n, d_tasks, d_results, d_record_pos, d_direction, d_records, d_count,
                             d_scratch, stream=None):
        _check(lib().wamd_emit_lev_records_dev(self.h, C.c_uint32(n), _ptr(d_tasks), _ptr(d_results),
                                               _ptr(d_record_pos), _ptr(d_direction), _ptr(d_records),
                                               _ptr(d_count), _ptr(d_scratch), C.c_void_p(stream or 0)))

    def emit_lev_records_static_dev(self, n, d_tasks, d_results, d_record_pos, d_direction, d_records, d_count,
                                    stream=None):
        """The no-alloc form: scratch in the library's static device array (n <= 2^24)."""
        _check(lib().wamd_emit_lev_records_static_dev(self.h, C.c_uint32(n), _ptr(d_tasks), _ptr(d_results),
                                                      _ptr(d_record_pos), _ptr(d_direction), _ptr(d_records),
                                                      _ptr(d_count), C.c_void_p(stream or 0)))

    def indel_match_dev(self, reads, d_tasks, n, d_results, params=None, stream=None):
        prm = params or default_indel_params()
        _check(lib().wamd_indel_match_dev(self.h, reads.h, _ptr(d_tasks), C.c_uint32(n), C.byref(prm),
                                          _ptr(d_results), C.c_void_p(stream or 0)))

    def align_workspace_bytes(self, reads, n, max_ed=None):
        if max_ed is None:
            return int(lib().wamd_align_workspace_bytes(reads.h, C.c_uint32(n)))
        return int(lib().wamd_align_workspace_bytes_for(reads.h, C.c_uint32(n), C.c_uint32(max_ed)))

    def align_dev(self, reads, d_tasks, n, d_results, d_cigars, cigar_stride, d_ws, ws_bytes,
                  last_symbol=0, refine=1, stream=None):
        _check(lib().wamd_align_dev(self.h, reads.h, _ptr(d_tasks), C.c_uint32(n),
                                    C.c_int(last_symbol), C.c_int(refine), _ptr(d_results),
                                    _ptr(d_cigars), C.c_uint32(cigar_stride), _ptr(d_ws),
                                    C.c_uint64(ws_bytes), C.c_void_p(stream or 0)))


def device_count():
    return int(lib().wamd_device_count())


def version():
    return lib().wamd_version().decode()


COMM_ID_BYTES = 128


def comm_unique_id():
    """ncclGetUniqueId through the C-ABI: 128 bytes rank 0 hands to the other ranks."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(lib().wamd_comm_get_unique_id(buf))
    return bytes(buf)


class Comm:
    """One rank's end of the record gather (wamd_comm_create)."""

    def __init__(self, ctx, rank, n_ranks, unique_id):
        self.ctx, self.rank, self.n_ranks = ctx, rank, n_ranks
        self.h = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(lib().wamd_comm_create(ctx.h, C.c_int(rank), C.c_int(n_ranks), buf, C.byref(self.h)))

    def gather_records_dev(self, d_records, d_count, capacity, d_out, root=0, record_bytes=12, stream=None):
        counts = np.zeros(self.n_ranks, np.uint32)
        _check(lib().wamd_gather_records_dev(self.h, _ptr(d_records), _ptr(d_count), C.c_uint32(record_bytes),
                                             C.c_uint32(capacity), C.c_int(root), _ptr(d_out), _ptr(counts),
                                             C.c_void_p(stream or 0)))
        return counts

    def destroy(self):
        if self.h:
            lib().wamd_comm_destroy(self.h)
            self.h = C.c_void_p()
