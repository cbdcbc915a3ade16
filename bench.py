#!/usr/bin/env python3
"""bench.py - verified candidates / second of the MI355X read-verification path.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the hot path over one batch of synthetic candidates that are
already resident in HBM: N_TASKS LevMyers256 distance verifications of shape S1
(m = 150, k = 12, W = 174 - the sensitive stages of the 150 bp configs,
BASELINE.json configs[1]).  For N > 1 every rank verifies its own shard (read bins
shard naturally) and the accepted candidates are gathered to rank 0 as
mapping records over RCCL - the path's only exchange step - inside the timed region.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the
stream the kernel runs on; `cpu_baseline` times the reference's own AVX2 verifier
(oracle/_ref, kind "reference") or, where that library is absent, our CPU restatement
(kind "port") on a bounded sample of the same tasks on this box's host cores.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from whisper_amd import api, gather, synth  # noqa: E402

M, K_ERR = 150, 12
W = M + 2 * K_ERR
DWORDS = (M - 1) // 32 + 1
N_COLS = W + 1 - (M - 1) // 64
# The op model of `roofline` is THE KERNEL'S OWN INSTRUCTION COUNT per lane and text column,
# i.e. `frac` is an issue-slot utilisation of the 2-cycle VALU peak, not a distance from an algorithmic optimum:
#   exact mode   myers_distance_kernel: 10 per 32-bit dword of the bit-vector + 20 of bookkeeping, over the
#                W + 1 - wn columns LevMyers256 iterates (LevMyers256.hpp:138)
#   accepted-only myers_band_kernel<2>: 2 dwords x 9 + 6 per sliding column, 2 x 10 + 24 per fixed column
# SURVEY 8(d)'s figure for the same task (40 lane-ops per 64-bit word-column + 8: 175 x 128 = 22 400) is reported
# beside it as `survey_8d`; on gfx950 it is not a lower bound (v_bitop3 and carry chains halve it), so its
# fraction can exceed 1.
LANE_OPS_EXACT = N_COLS * (DWORDS * 10 + 20)
BAND_FS = (N_COLS + K_ERR - 64 + 7) // 8 * 8
LANE_OPS_BAND = BAND_FS * 24 + (N_COLS - BAND_FS) * 44
LANE_OPS_SURVEY_8D = (W + 1) * (((M - 1) // 64 + 1) * 40 + 8)
# `roofline.algorithmic`: the recurrence alone - the ten (exact) or nine / ten (band) operations per 32-bit dword and
# column that the Myers / Hyyro update cannot do without on this ISA, none of the kernel's bookkeeping (minimum
# tracking, symbol fetch, loop control).  `frac` against this count is the distance from an algorithmic bound; `frac`
# of the instruction-count model above is issue-slot utilisation.
LANE_OPS_EXACT_RECURRENCE = N_COLS * DWORDS * 10
LANE_OPS_BAND_RECURRENCE = BAND_FS * 2 * 9 + (N_COLS - BAND_FS) * 2 * 10
# algorithmic bytes per task: 2-bit read + 2-bit window + 16 B task + 4 B result
BYTES_PER_TASK = (M + 3) // 4 + (W + 3 + 3) // 4 + 16 + 4
# MI355X: 256 CUs x 4 SIMD x 32 lanes/clk x 2.4 GHz (MI355X_MICROARCH.md: SIMD-32, 2400 MHz)
PEAK_VALU_LANEOPS = 256 * 4 * 32 * 2.4e9
PEAK_HBM = 8.0e12


def roof_valu(units, ops_per_unit, ms, kernel=None):
    a = units * ops_per_unit / (ms * 1e-3)
    r = {"bound": "valu", "achieved": a / 1e12, "peak": PEAK_VALU_LANEOPS / 1e12,
         "unit": "T lane-op/s (32-bit integer VALU)", "frac": a / PEAK_VALU_LANEOPS, "traffic": None,
         "model": "kernel instruction count: %d lane-ops per unit" % ops_per_unit}
    if kernel:
        r["kernel"] = kernel
    return r


def roof_hbm(units, bytes_per_unit, ms, kernel=None):
    a = units * bytes_per_unit / (ms * 1e-3)
    r = {"bound": "hbm", "achieved": a / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": a / PEAK_HBM,
         "traffic": None, "algorithmic_bytes_per_unit": bytes_per_unit}
    if kernel:
        r["kernel"] = kernel
    return r


def measured_counters():
    """Per-kernel PMC figures of the committed profile of this tree, if any (profiles/round3_counters.json):
    HBM bytes per unit and VALU instructions per unit, keyed by the extras / roofline tag."""
    for name in ("round3_counters.json", "round2_counters.json"):
        try:
            return json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
    return {}


# ---- profile markers -------------------------------------------------------------------------------------------
# `--profile-markers` brackets every timed region with two tiny float64 fill kernels of distinct sizes, so that a
# rocprofv3 counter pass (which cannot carry marker traces together with --pmc on this pool) can still be cut into
# "the dispatches of S2", "of S4", ...: scripts/summarize_extras.py ranks the FillFunctor<double> dispatches by grid
# size (rank 2i = begin of region i, 2i + 1 = its end; the regions in the order of MARKS["tags"]).
MARKS = {"on": False, "tags": [], "reps": {}, "units": {}}


def mark(tag, begin, stream, device, reps=None, units=None):
    if not MARKS["on"]:
        return
    if begin:
        MARKS["tags"].append(tag)
        MARKS["reps"][tag] = reps
        MARKS["units"][tag] = units
    i = 2 * MARKS["tags"].index(tag) + (0 if begin else 1)
    with torch.cuda.stream(stream):
        torch.empty((i + 1) * 65536, dtype=torch.float64, device=device).fill_(1.0)


def host_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"cpu_model": model, "os_cpu_count": os.cpu_count()}


def env_int(name, default):
    try:
        return int(os.environ.get(name, default))
    except ValueError:
        return default


def build_workload(device, n_tasks, genome_symbols, seed, true_frac=0.5):
    """Genome, reads and tasks of shape S1, generated on the GPU; returns host copies of
    what the upload entry points take plus the device task array."""
    codes = synth.make_genome(genome_symbols, seed=0xC0FFEE, device=device)
    packed = synth.pack_nibbles(codes)
    rd = synth.make_reads(codes, n_tasks, M, K_ERR, seed, true_frac=true_frac, chunk=1 << 17)
    blob, off, lens = synth.pack_reads(rd["codes"], 0)
    tasks = synth.distance_tasks(rd, M, K_ERR)
    ref_size = codes.numel()
    del codes, rd
    return packed, ref_size, blob.cpu().numpy(), off.cpu().numpy().astype(np.uint64), \
        lens.cpu().numpy().astype(np.uint32), tasks


CANDIDATES_PER_READ = 4   # c: SURVEY 3.4 implies 4-8 SA entries per anchor on a repetitive genome


def cpu_baseline(packed_host, ref_size, blob, off, lens, tasks, sample, threads):
    """Reference (or port) on the host cores, bounded sample of the same tasks.  The step's workload has one read
    per candidate, so `value` charges the CPU one preprocess() per candidate; the reference preprocesses once per
    read (mapping_core_sa.cpp:212-213).  preprocess() is therefore also timed alone and the rate a candidate sees
    when its read's table is shared by c = CANDIDATES_PER_READ candidates is reported next to it."""
    from oracle import bindings
    t = tasks[:sample]
    info = host_info()
    if bindings.have_reference():
        try:
            ref = bindings.Reference()
            kw = dict(max_query_len=255, max_text_len=M + 2 * K_ERR + 2, nthreads=threads)
            ref.distance(packed_host, ref_size, blob, off, lens, 0, t[:4096], -1, **kw)
            t0 = time.perf_counter()
            rc, res = ref.distance(packed_host, ref_size, blob, off, lens, 0, t, -1, **kw)
            dt = time.perf_counter() - t0
            assert rc == 0
            n_pre = int(t["read_idx"].max()) + 1
            t0 = time.perf_counter()
            ref.preprocess_only(blob, off, lens[:n_pre], 0, **kw)
            dt_pre = time.perf_counter() - t0
            dp_only = len(t) / max(dt - dt_pre, 1e-9)
            c = CANDIDATES_PER_READ
            at_c = len(t) / (max(dt - dt_pre, 1e-9) + dt_pre / c)
            return dict(value=len(t) / dt, unit="candidates/s", cores=threads, kind="reference",
                        sample="first %d of the step's S1 tasks, LevMyers256<avx2> preprocess + "
                               "dynamicProgramming per task, %d threads, %.2f s" % (len(t), threads, dt),
                        preprocess_reads_per_s=n_pre / dt_pre, dynamic_programming_only_per_s=dp_only,
                        candidates_per_read_c=c, value_at_c_candidates_per_read=at_c, host=info), res
        except Exception as e:  # e.g. no AVX2 on this host
            sys.stderr.write("reference baseline unavailable (%s); timing the port\n" % e)
    from concurrent.futures import ThreadPoolExecutor
    orc = bindings.Oracle()
    t = t[: max(1, sample // 4)]
    chunks = np.array_split(np.arange(len(t)), threads * 4)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(lambda ix: orc.distance(packed_host, blob, off, lens, 0, t[ix], -1)[1], chunks))
    dt = time.perf_counter() - t0
    return dict(value=len(t) / dt, unit="candidates/s", cores=threads, kind="port",
                sample="first %d of the step's S1 tasks, scalar C restatement, %d threads, %.2f s"
                       % (len(t), threads, dt), host=info), np.concatenate(parts)


def boundary_costs(ctx, device, blob, off, lens, tasks, n):
    """What the boundary costs when it hands over HOST buffers (never `value`): the upload of the reads
    (wamd_reads_upload: H2D + the prepare kernels, per read and orientation pair), the synchronous host-pointer
    call, and a two-stream pipeline of the _dev entry point over pinned buffers (H2D of 16 B tasks, kernel, D2H of
    8 B results overlapped chunk by chunk)."""
    out = {}
    t0 = time.perf_counter()
    reads = ctx.upload_reads(blob, off, lens, api.READS_NIBBLE, orientations=3)
    ctx.synchronize()
    up = time.perf_counter() - t0
    out["reads_upload_ms"] = up * 1e3
    out["reads_upload_ns_per_read"] = up / len(lens) * 1e9
    ctx.dynamic_programming(reads, tasks[:1 << 16], -1)          # staging buffers grow once
    t0 = time.perf_counter()
    res = ctx.dynamic_programming(reads, tasks, -1)
    dt = time.perf_counter() - t0
    out["host_pointer_call_candidates_per_s"] = n / dt
    # pipeline: pinned host tasks / results, chunks alternate between two streams
    h_tasks = torch.from_numpy(tasks.view(np.uint8).reshape(n, 16)).pin_memory()
    h_res = torch.empty((n, 8), dtype=torch.uint8).pin_memory()
    n_chunks = 16
    bounds = np.linspace(0, n, n_chunks + 1).astype(np.int64)
    cmax = int(np.diff(bounds).max())
    streams = [torch.cuda.Stream(device=device) for _ in range(2)]
    d_t = [torch.empty((cmax, 16), dtype=torch.uint8, device=device) for _ in range(2)]
    d_r = [torch.empty((cmax, 8), dtype=torch.uint8, device=device) for _ in range(2)]

    def run():
        for ci in range(n_chunks):
            a, b = int(bounds[ci]), int(bounds[ci + 1])
            st = streams[ci & 1]
            with torch.cuda.stream(st):
                d_t[ci & 1][: b - a].copy_(h_tasks[a:b], non_blocking=True)
                ctx.dynamic_programming_dev(reads, d_t[ci & 1], b - a, d_r[ci & 1], last_symbol=-1, stream=st.cuda_stream)
                h_res[a:b].copy_(d_r[ci & 1][: b - a], non_blocking=True)
        for st in streams:
            st.synchronize()

    run()
    t0 = time.perf_counter()
    run()
    dt = time.perf_counter() - t0
    out["host_to_host_two_streams_candidates_per_s"] = n / dt
    out["host_to_host_matches_sync_call"] = bool((h_res.numpy().view(api.DIST_RESULT).reshape(-1) == res).all())
    out["bytes_over_pcie_per_candidate"] = 24
    reads.free()
    return out


def boundary_compact(ctx, device, blob, off, lens, tasks, n, ref_size):
    """The boundary as the host mapper would use it (never `value`): the suffix-array loops' compact submission form
    (wamd_verify_sa_candidates: per read two 2-byte counts, per candidate its 4-byte suffix-array entry in, one byte
    of errors out) for C candidates per read - each read's own S1 locus and C - 1 random ones, on the read's strand -
    in bins of 2^19 reads, host buffers in, host buffers out.  Two rates: the bins' reads already resident, and the
    reads of bin i + 1 uploaded by a second host thread (wamd_reads_upload_stream) beside the kernels of bin i, which is
    what a mapper that fills the next bin while this one is verified sees."""
    import threading
    c = 8
    k = K_ERR
    fl, fr = synth.stage_segments(M, k)[1]
    n_reads = n // c
    bins = int(os.environ.get("WAMD_BENCH_BINS", "4"))
    per = n_reads // bins
    rng = np.random.default_rng(1234)
    out = {"candidates_per_read": c, "reads_per_bin": per, "bins": bins,
           "form": "wamd_sa_batch + counts (2 x u16 per read) + SA entries (u32 per candidate) -> errors (u8 per candidate)"}
    t = tasks[:n_reads]
    rc = (t["flags"] & 1).astype(bool)
    m = lens[:n_reads].astype(np.int64)
    # the window start the S1 task has -> the suffix-array entry the loop would have had for it (mapping_core_sa.cpp:237, 289)
    ref_pos = np.empty((n_reads, c), np.int64)
    ref_pos[:, 0] = t["ref_pos"]
    ref_pos[:, 1:] = rng.integers(1 << 16, ref_size - (1 << 16), (n_reads, c - 1))
    sa = np.where(rc[:, None], ref_size - m[:, None] + fl - k - ref_pos, ref_pos + k + fl).astype(np.uint32)
    counts = np.zeros((n_reads, 2), np.uint16)
    counts[:, 0] = np.where(rc, 0, c)
    counts[:, 1] = np.where(rc, c, 0)
    p_counts, p_sa, p_err = api.PinnedBuffer(2 * n_reads, np.uint16), api.PinnedBuffer(n_reads * c, np.uint32), \
        api.PinnedBuffer(n_reads * c, np.uint8)
    p_counts.array[:] = counts.reshape(-1)
    p_sa.array[:] = sa.reshape(-1)
    o0, o1 = int(off[0]), int(off[n_reads - 1]) + (int(lens[n_reads - 1]) + 1) // 2
    p_blob, p_off, p_lens = api.PinnedBuffer(o1 - o0 + 64, np.uint8), api.PinnedBuffer(n_reads, np.uint64), \
        api.PinnedBuffer(n_reads, np.uint32)
    p_blob.array[:o1 - o0] = blob[o0:o1]
    p_lens.array[:] = lens[:n_reads]
    bin_base = [int(off[b * per]) - o0 for b in range(bins)]
    for b in range(bins):   # a bin's offsets count from its own first read
        p_off.array[b * per:(b + 1) * per] = off[b * per:(b + 1) * per] - off[b * per]

    def upload(b, stream=None):
        a, e = b * per, (b + 1) * per
        return ctx.upload_reads(p_blob.array[bin_base[b]:], p_off.array[a:e], p_lens.array[a:e], api.READS_NIBBLE,
                                orientations=3, stream=stream)

    def verify(b, reads):
        a, e = b * per, (b + 1) * per
        ctx.verify_sa_candidates(reads, api.sa_batch(per, k, (fl, fr), 0, api.SA_VERIFIER_DISTANCE),
                                 p_counts.array[2 * a:2 * e], p_sa.array[a * c:e * c], errors=p_err.array[a * c:e * c])

    # (1) reads resident
    resident = [upload(b) for b in range(bins)]
    for b in range(bins):                       # staging buffers and streams exist after this, and every stream has
        verify(b, resident[b])                  # copied from every buffer once (the first such copy is slow)
    def median_pass(fn, passes=5):
        # a pass over the bins is a few milliseconds of host-driven work: one stall of the host shows in it, so the
        # rate is the median of five passes
        ts = []
        for _ in range(passes):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    def resident_pass():
        for b in range(bins):
            verify(b, resident[b])

    dt = median_pass(resident_pass)
    out["host_to_host_compact_candidates_per_s"] = bins * per * c / dt
    # parity: the task form of bin 0's candidates through the synchronous host-pointer call, clamped at k + 1
    tt = np.zeros(per * c, api.DIST_TASK)
    tt["read_idx"] = np.repeat(np.arange(per, dtype=np.uint32), c)
    tt["ref_pos"] = ref_pos[:per].reshape(-1).astype(np.uint32)
    tt["max_distance_in_ref"] = np.repeat(m[:per] + 2 * k, c)
    tt["max_ed"] = k
    tt["flags"] = np.repeat(rc[:per].astype(np.uint8), c)
    want = ctx.dynamic_programming(resident[0], tt, -1)["edit_distance"]
    out["compact_matches_task_form"] = bool((np.minimum(want, k + 1) == p_err.array[:per * c]).all())
    out["compact_accepted_fraction"] = float((p_err.array[:per * c] <= k).mean())
    for r in resident:
        r.free()
    # (2) the next bin's reads uploaded beside this bin's kernels, into one of two batches that take turns
    # (wamd_reads_refill: after its first two bins a mapper allocates nothing)
    up_stream = torch.cuda.Stream(device=device)
    batch = [upload(0, up_stream.cuda_stream), upload(1, up_stream.cuda_stream)]

    def refill(b):
        a, e = b * per, (b + 1) * per
        batch[b & 1].refill(p_blob.array[bin_base[b]:], p_off.array[a:e], p_lens.array[a:e], stream=up_stream.cuda_stream)

    # one worker thread fills the other batch while this thread verifies (what a mapper's reader thread would do)
    import queue
    todo, done = queue.Queue(), queue.Queue()

    def worker():
        while True:
            b = todo.get()
            if b is None:
                return
            try:
                refill(b)
                done.put(None)
            except BaseException as e:     # hand the failure to the timing thread instead of dying silently
                done.put(e)

    th = threading.Thread(target=worker, daemon=True)
    th.start()

    def pipeline():
        refill(0)
        for b in range(bins):
            if b + 1 < bins:
                todo.put(b + 1)
            verify(b, batch[b & 1])
            if b + 1 < bins:
                e = done.get()
                if e is not None:
                    raise e

    pipeline()      # (the first pass pays for the second thread's first calls into the runtime)
    dt = median_pass(pipeline)
    todo.put(None)
    th.join()
    out["host_to_host_compact_with_reads_upload_candidates_per_s"] = bins * per * c / dt
    # the same with wamd_reads_refill_async (sizes stated, no host pass, no wait): one host thread enqueues bin b + 1's
    # reads on the upload stream, verifies bin b, and waits for the upload stream before it goes on
    max_len = int(lens[:n_reads].max())
    bin_bytes = [(int(off[(b + 1) * per - 1]) + (int(lens[(b + 1) * per - 1]) + 1) // 2) - int(off[b * per]) for b in range(bins)]

    def refill_async(b):
        a, e = b * per, (b + 1) * per
        batch[b & 1].refill_async(p_blob.array[bin_base[b]:], bin_bytes[b], p_off.array[a:e], p_lens.array[a:e], max_len,
                                  stream=up_stream.cuda_stream)

    def pipeline_async():
        refill_async(0)
        up_stream.synchronize()
        for b in range(bins):
            if b + 1 < bins:
                refill_async(b + 1)
            verify(b, batch[b & 1])
            up_stream.synchronize()

    pipeline_async()
    pipeline_async()      # (two passes before the clock: the first calls on a new path are slow twice over)
    p_err.array[:] = 0xEE
    dt = median_pass(pipeline_async)
    out["host_to_host_compact_with_async_reads_refill_candidates_per_s"] = bins * per * c / dt
    out["async_refill_matches_task_form"] = bool((np.minimum(want, k + 1) == p_err.array[:per * c]).all()) and \
        ctx.take_flags(up_stream.cuda_stream) == 0
    t0 = time.perf_counter()
    refill_async(0)
    out["reads_refill_async_enqueue_ms_per_bin"] = (time.perf_counter() - t0) * 1e3
    up_stream.synchronize()
    out["reads_refill_async_ms_per_bin"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    refill(0)
    out["reads_refill_ms_per_bin"] = (time.perf_counter() - t0) * 1e3
    for r in batch:
        r.free()
    out["bytes_over_pcie_per_candidate"] = 4 + 1 + 4 / c
    out["reads_bytes_over_pcie_per_candidate"] = float((o1 - o0 + 12 * n_reads) / (n_reads * c))
    for b in (p_counts, p_sa, p_err, p_blob, p_off, p_lens):
        b.free()
    return out


def run_extras(ctx, device, genome_symbols, stream, n_small, gather_cabi=False):
    """Secondary shapes of SURVEY 8d and the rows next to the verifier on the same resident genome: one entry per
    kernel family, timed with HIP events on the context's stream (3 launches after one warm-up), each with the
    roofline that bounds it.  Algorithmic bytes per unit are SURVEY 8(d)'s accounting: 2-bit read + 2-bit window +
    task + result (+ the ext-CIGAR bytes a record scan reads and writes)."""
    codes = synth.make_genome(genome_symbols, seed=0xC0FFEE, device=device)
    G = codes.numel()
    out = {}

    def timed(fn, reps=3, prep=None, tag=None, units=None):
        if prep:
            prep()
        fn()
        ctx.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        if tag:
            mark(tag, True, stream, device, reps, units)
        # (a collection of Python's garbage between an event and the launch behind it once put 36 ms of host stall
        #  into a 2.7 ms kernel's bracket: collect before, not during; and report the median bracket)
        gc.collect()
        gc.disable()
        try:
            for a, b in evs:
                if prep:
                    prep()
                a.record(stream)
                fn()
                b.record(stream)
        finally:
            gc.enable()
        if tag:
            mark(tag, False, stream, device)
        ctx.synchronize()
        f = ctx.take_flags()
        if f:
            raise SystemExit("device error flags 0x%x in extras" % f)
        return float(np.median([a.elapsed_time(b) for a, b in evs]))

    def host(rd_codes, enc):
        blob, off, lens = synth.pack_reads(rd_codes, enc)
        return blob.cpu().numpy(), off.cpu().numpy().astype(np.uint64), lens.cpu().numpy().astype(np.uint32)

    def dev_bytes(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(device)

    def q(x):          # 2-bit bytes of x symbols
        return (x + 3) // 4

    # S2: LevDiag, m = 150, k = 5, anchor = segment 2 of stage 5 (mapping_core_sa.cpp:239)
    n = n_small
    fl, fr = synth.stage_segments(150, 5)[2]
    rd = synth.make_reads(codes, n, 150, 5, 77, rc_frac=0.0, protect=(fl, fr), chunk=1 << 17)
    reads = ctx.upload_reads(*host(rd["codes"], 0), api.READS_NIBBLE, orientations=1)
    t = synth.levdiag_tasks(rd, 5, fl, fr)
    d_t = dev_bytes(t)
    d_r = torch.zeros(n, dtype=torch.int32, device=device)
    ms = timed(lambda: ctx.lev_diag_dev(reads, d_t, n, d_r), tag="S2_levdiag", units=n)
    acc = float((d_r.cpu().numpy().view(np.uint32) <= 5).mean())
    out["S2_levdiag"] = {"shape": "CEditDist::LevDiag m=150 k=5 anchor [%d,%d)" % (fl, fr), "tasks": n,
                         "kernel_ms": ms, "candidates_per_s": n / ms * 1e3, "accepted_fraction": acc,
                         # per column of either side's band: 6 ops of match vector, 9 of recurrence, 2 of score
                         "roofline": roof_valu(n, ((fl + 5) + (150 - fr + 5)) * 17, ms, "levdiag_band_kernel<5,256>"),
                         "hbm_bytes_per_candidate": 80 + q(150 + 2 * 5 + 1) + 16 + 4}
    reads.free()
    del d_t, d_r

    # CF pre-filter in front of S2: same reads, anchor and loci (mapping_core_sa.cpp:228-230)
    tc = np.zeros(n, api.CF_TASK)
    tc["read_idx"] = np.arange(n, dtype=np.uint32)
    tc["position"] = rd["p"].cpu().numpy().astype(np.uint32)
    tc["pattern_len"] = 150
    tc["fixed_left"], tc["fixed_right"] = fl, fr
    tc["dir_gen_flag"] = 1
    d_tc = dev_bytes(tc)
    d_rc = torch.zeros(n, dtype=torch.int32, device=device)
    reads = ctx.upload_reads(*host(rd["codes"], 0), api.READS_NIBBLE, orientations=1)
    ms = timed(lambda: ctx.cf_filter_dev(reads, d_tc, n, d_rc), tag="C_cf_filter", units=n)
    out["C_cf_filter"] = {"shape": "CF pre-filter m=150 anchor [%d,%d), pass iff L1 <= 20" % (fl, fr), "tasks": n,
                          "kernel_ms": ms, "candidates_per_s": n / ms * 1e3,
                          "pass_fraction": float((d_rc.cpu().numpy().view(np.uint32) <= 20).mean()),
                          "roofline": roof_hbm(n, 2 * q(150) + 16 + 4, ms, "cf_filter_kernel")}
    # the loop body as one call: filter, survivors packed into full waves, LevDiag on those (mapping_core_sa.cpp:223-255)
    d_t = dev_bytes(t)
    d_r = torch.zeros(n, dtype=torch.int32, device=device)
    d_r2 = torch.zeros(n, dtype=torch.int32, device=device)
    fwb = ctx.filter_verify_workspace_bytes(n)
    d_fw = torch.zeros(fwb + 256, dtype=torch.uint8, device=device)
    padf = (-d_fw.data_ptr()) % 256
    ms_f = timed(lambda: ctx.filter_levdiag_dev(reads, d_tc, d_t, n, d_r2, d_fw[padf:], fwb), tag="F_filter_levdiag", units=n)
    ctx.lev_diag_dev(reads, d_t, n, d_r)
    ctx.synchronize()
    passed = d_rc.cpu().numpy().view(np.uint32) <= 20
    alone, fused = d_r.cpu().numpy().view(np.uint32), d_r2.cpu().numpy().view(np.uint32)
    out["F_filter_levdiag"] = {"shape": "CF pre-filter -> pack -> LevDiag on the survivors -> unpack, S2's candidates, one call",
                               "tasks": n, "total_ms": ms_f, "candidates_per_s": n / ms_f * 1e3,
                               "separate_calls_ms": out["C_cf_filter"]["kernel_ms"] + out["S2_levdiag"]["kernel_ms"],
                               "same_as_separate_calls": bool((fused == np.where(passed, alone, 6)).all()),
                               "roofline": roof_hbm(n, 2 * q(150) + 16 + 16 + 4 + float(passed.mean()) * (80 + q(161) + 2 * 20), ms_f,
                                                    "filter_pack_kernel + levdiag_band_kernel + filter_unpack_kernel")}
    reads.free()
    del d_tc, d_rc, d_t, d_r, d_r2, d_fw

    # S5: LevMyers256 distance, m = 250, k = 17, W = 284 (8 dwords per vector)
    rd = synth.make_reads(codes, n, 250, 17, 78, chunk=1 << 16)
    reads = ctx.upload_reads(*host(rd["codes"], 0), api.READS_NIBBLE, orientations=3)
    t = synth.distance_tasks(rd, 250, 17)
    d_t = dev_bytes(t)
    d_r = torch.zeros(n * 2, dtype=torch.int32, device=device)
    ctx.set_max_text_len(16384)
    ms = timed(lambda: ctx.dynamic_programming_dev(reads, d_t, n, d_r, last_symbol=-1), tag="S5_distance_250bp", units=n)
    cols5 = 284 + 1 - 3
    out["S5_distance_250bp"] = {"shape": "LevMyers256 distance m=250 k=17 W=284", "tasks": n, "kernel_ms": ms,
                                "candidates_per_s": n / ms * 1e3,
                                "roofline": roof_valu(n, cols5 * (8 * 10 + 20), ms, "myers_distance_kernel<8,256>")}
    ctx.set_distance_mode(api.DISTANCE_ACCEPTED_ONLY)
    ms = timed(lambda: ctx.dynamic_programming_dev(reads, d_t, n, d_r, last_symbol=-1), tag="S5_accepted_only", units=n)
    ctx.set_distance_mode(api.DISTANCE_EXACT)
    fs5 = (cols5 + 17 - 96 + 7) // 8 * 8
    out["S5_accepted_only"] = {"shape": "S5 in WAMD_DISTANCE_ACCEPTED_ONLY mode (96-row band)", "tasks": n,
                               "kernel_ms": ms, "candidates_per_s": n / ms * 1e3,
                               "roofline": roof_valu(n, fs5 * (3 * 9 + 6) + (cols5 - fs5) * (3 * 10 + 24), ms,
                                                     "myers_band_kernel<3,256,true> (after <2,256,true> deferred everything)")}
    # K: the accepted candidates of this batch as 12-byte mapping records in task order (results.cpp:161-173)
    d_pos = torch.from_numpy((t["ref_pos"] + 17).astype(np.uint32).view(np.int32)).to(device)
    d_dir = torch.ones(n, dtype=torch.uint8, device=device)
    d_rec = torch.zeros((n, 12), dtype=torch.uint8, device=device)
    d_cnt = torch.zeros(1, dtype=torch.int32, device=device)
    ctx.dynamic_programming_dev(reads, d_t, n, d_r, last_symbol=-1)
    ms = timed(lambda: ctx.emit_lev_records_static_dev(n, d_t, d_r, d_pos, d_dir, d_rec, d_cnt), tag="K_emit_lev_records", units=n)
    n_acc = int(d_cnt.item())
    out["K_emit_lev_records"] = {"shape": "accepted candidates -> 12-byte records in task order (count, scan, write)",
                                 "tasks": n, "records": n_acc, "kernel_ms": ms, "candidates_per_s": n / ms * 1e3,
                                 "roofline": roof_hbm(n, 2 * (16 + 8) + (4 + 1 + 12) * n_acc / n, ms,
                                                      "records_count + records_scan + records_write")}
    reads.free()
    del d_t, d_r, d_pos, d_dir, d_rec, d_cnt

    # S4: fill_cigar_with_lev: distance + traceback + ext-CIGAR + refinement, m = 150, W = 177
    na = n // 4
    rd = synth.make_reads(codes, na, 150, 10, 79, true_frac=1.0, chunk=1 << 16)
    reads = ctx.upload_reads(*host(rd["codes"], 1), api.READS_ASCII, orientations=3)
    t = synth.distance_tasks(rd, 150, 13, W=177)
    t["ref_pos"] = (rd["p"] - 13).cpu().numpy().astype(np.uint32)
    d_t = dev_bytes(t)
    stride = 512
    d_r = torch.zeros(na * api.ALIGN_RESULT.itemsize, dtype=torch.uint8, device=device)
    d_c = torch.zeros(na * stride, dtype=torch.uint8, device=device)
    wsb = ctx.align_workspace_bytes(reads, na)
    d_w = torch.zeros(wsb, dtype=torch.uint8, device=device)
    ms = timed(lambda: ctx.align_dev(reads, d_t, na, d_r, d_c, stride, d_w, wsb, last_symbol=0, refine=1), tag="S4_align", units=na)
    res = d_r.cpu().numpy().view(api.ALIGN_RESULT)
    clen = float(res["cigar_len"][res["ok"] == 1].mean())
    out["S4_align"] = {"shape": "dynamicProgramming + getExtCigar + refine_ext_cigar m=150 k=13 W=177", "tasks": na,
                       "total_ms": ms, "alignments_per_s": na / ms * 1e3, "ok_fraction": float(res["ok"].mean()),
                       "workspace_bytes": wsb, "mean_ext_cigar_bytes": clen,
                       "roofline": roof_hbm(na, q(150) + q(180) + 16 + 40 + clen, ms,
                                            "myers_distance + align_classify + align_traceback x2")}
    # the same call with the context in accepted-only mode: pass 1 is the banded kernel (no whole-column pass for
    # tasks whose band fits); what the reference goes on with - the accepted tasks - must come back identical
    cig = d_c.view(na, stride).cpu().numpy()
    ctx.set_distance_mode(api.DISTANCE_ACCEPTED_ONLY)
    ms_b = timed(lambda: ctx.align_dev(reads, d_t, na, d_r, d_c, stride, d_w, wsb, last_symbol=0, refine=1),
                 tag="S4_align_accepted_only", units=na)
    ctx.set_distance_mode(api.DISTANCE_EXACT)
    res_b = d_r.cpu().numpy().view(api.ALIGN_RESULT)
    cig_b = d_c.view(na, stride).cpu().numpy()
    acc = res["ok"] == 1
    same = bool((res_b[acc] == res[acc]).all() and (res_b["ok"][~acc] == 0).all())
    for i in np.flatnonzero(acc)[:: max(1, int(acc.sum()) // 4096)]:   # strings up to their terminator, a sample
        e = int(res["cigar_len"][i])
        same = same and bool((cig_b[i, :e + 1] == cig[i, :e + 1]).all())
    out["S4_align_accepted_only"] = {"shape": "S4 with the banded first pass (WAMD_DISTANCE_ACCEPTED_ONLY)", "tasks": na,
                                     "total_ms": ms_b, "alignments_per_s": na / ms_b * 1e3,
                                     "accepted_identical": same,
                                     "roofline": roof_hbm(na, q(150) + q(180) + 16 + 40 + clen, ms_b,
                                                          "myers_band + align_classify + align_traceback x2")}
    del cig, cig_b, res_b
    reads.free()
    del d_t, d_r, d_w

    # Rows B, L, X on the ext-CIGARs S4 just produced, as fill_cigar_with_lev leaves the records
    # (sam_matching.cpp:737-746): every fourth read hangs over an end of its chromosome, pairs are made of
    # neighbouring slots with the reverse read on the left
    rec = np.zeros(na, api.CLIP_RECORD)
    rec["score"] = res["affine_score"]
    rec["ref_seq_pos"] = 1000 + res["begin_pos"].astype(np.int64)
    rec["ref_length"] = np.where(res["ok"] == 1, res["end_pos"] - res["begin_pos"], 0)
    rec["err_edit_distance"] = np.where(res["ok"] == 1, res["edit_distance"], 0)
    ti = np.zeros(na, api.CLIP_ILLEGAL_TASK)
    ti["read_length"] = 150
    okr = res["ok"] == 1
    hang = (np.arange(na) % 4 == 0) & okr
    rec["ref_seq_pos"][hang & (np.arange(na) % 8 == 0)] = -7
    ti["ref_size"] = np.where(hang & (np.arange(na) % 8 == 4), rec["ref_seq_pos"] + 120, 1 << 30)
    tp = np.zeros(na // 2, api.CLIP_PAIR_TASK)
    tp["ref_seq_id"][:, 0] = 1
    tp["ref_seq_id"][:, 1] = np.where(okr[0::2] & okr[1::2], 1, 2)     # a pair with an unaligned read is left alone
    tp["read_length"] = 150
    tp["direct"][:, 1] = 1
    rec_pairs = rec.copy()
    rec_pairs["ref_seq_pos"][0::2] = 5000
    rec_pairs["ref_seq_pos"][1::2] = 5000 + (np.arange(na // 2) % 100)
    d_rec0, d_recp0, d_cig0 = dev_bytes(rec), dev_bytes(rec_pairs), d_c.clone()
    d_rec, d_ti, d_tp = torch.empty_like(d_rec0), dev_bytes(ti), dev_bytes(tp)
    scan = 2 * (clen + 32)          # an ext-CIGAR and its record, read and written back

    def fresh(src):
        def go():
            with torch.cuda.stream(stream):
                d_rec.copy_(src, non_blocking=True)
                d_c.copy_(d_cig0, non_blocking=True)
        return go

    ms = timed(lambda: ctx.clip_illegal_dev(na, d_ti, d_rec, d_c, stride), prep=fresh(d_rec0), tag="L_clip_illegal", units=na)
    out["L_clip_illegal"] = {"shape": "clipIllegalPositions on S4's ext-CIGARs, 1 in 4 over a chromosome end",
                             "records": na, "kernel_ms": ms, "records_per_s": na / ms * 1e3,
                             "roofline": roof_hbm(na, scan + 8, ms, "clip_illegal_kernel")}
    ms = timed(lambda: ctx.clip_boundary_dev(na, d_rec, d_c, stride), prep=fresh(d_rec0), tag="B_clip_boundary", units=na)
    out["B_clip_boundary"] = {"shape": "clipBoundaryPositions on S4's ext-CIGARs", "records": na, "kernel_ms": ms,
                              "records_per_s": na / ms * 1e3, "roofline": roof_hbm(na, scan, ms, "clip_boundary_kernel")}
    ms = timed(lambda: ctx.clip_overlapping_pairs_dev(na // 2, d_tp, d_rec, d_c, stride), prep=fresh(d_recp0),
               tag="L_clip_overlapping_pairs", units=na // 2)
    out["L_clip_overlapping_pairs"] = {"shape": "clipOverlappingPairs, reverse read left of its mate by 0..99", "pairs": na // 2,
                                       "kernel_ms": ms, "pairs_per_s": na // 2 / ms * 1e3,
                                       "roofline": roof_hbm(na // 2, 2 * scan + 16, ms, "clip_pairs_kernel")}
    d_sc = torch.zeros(na * 336, dtype=torch.uint8, device=device)
    d_sm = torch.zeros(na * 336, dtype=torch.uint8, device=device)
    d_sl = torch.zeros(na * 16, dtype=torch.uint8, device=device)
    fresh(d_rec0)()
    ms = timed(lambda: ctx.sam_cigar_dev(na, d_c, stride, d_sc, 336, d_sm, 336, d_sl), tag="X_sam_cigar", units=na)
    sl = d_sl.cpu().numpy().view(api.SAM_CIGAR_LEN)
    out["X_sam_cigar"] = {"shape": "extract_cigar_and_mdz on S4's ext-CIGARs", "records": na, "kernel_ms": ms,
                          "records_per_s": na / ms * 1e3,
                          "roofline": roof_hbm(na, clen + float(sl["cigar_len"].mean() + sl["mdz_len"].mean()) + 16, ms,
                                               "sam_cigar_kernel")}
    del d_rec0, d_recp0, d_cig0, d_rec, d_ti, d_tp, d_sc, d_sm, d_sl, d_c

    # The same rows as ONE call on mapping records (sam.cpp:1871-1900, 1992-2007): pairs of neighbouring reads,
    # lev records carrying the distance S4 found, perfect matches where it is 0
    np_pairs = na // 2
    ft = np.zeros(na, api.FINISH_TASK)
    ft["read_idx"] = np.arange(na, dtype=np.uint32)
    is_rc = rd["is_rc"].cpu().numpy()
    p0 = rd["p"].cpu().numpy().astype(np.int64)
    ft["direct"] = ~is_rc
    ft["pos"] = np.where(is_rc, p0 + 149, p0)
    ft["err_edit_distance"] = np.where(res["ok"] == 1, res["edit_distance"], 13)
    ft["ref_seq_pos"] = p0 + 1
    ft["ref_size"] = G
    d_ft = dev_bytes(ft)
    d_fr = torch.zeros(na * api.FINISH_RESULT.itemsize, dtype=torch.uint8, device=device)
    d_fe = torch.zeros(na * 352, dtype=torch.uint8, device=device)
    d_fc = torch.zeros(na * 352, dtype=torch.uint8, device=device)
    d_fm = torch.zeros(na * 352, dtype=torch.uint8, device=device)
    reads_fin = ctx.upload_reads(*host(rd["codes"], 1), api.READS_ASCII, orientations=3)
    fwb = ctx.finish_workspace_bytes(reads_fin, np_pairs)
    d_fw = torch.zeros(fwb + 256, dtype=torch.uint8, device=device)
    padw = (-d_fw.data_ptr()) % 256
    prm = api.default_finish_params(14)
    ms = timed(lambda: ctx.finish_pairs_dev(reads_fin, d_ft, np_pairs, d_fr, d_fe, 352, d_fc, 352, d_fm, 352, d_fw[padw:],
                                            fwb, params=prm), tag="N1_finish_pairs", units=np_pairs)
    fres = d_fr.cpu().numpy().view(api.FINISH_RESULT)
    out["N1_finish_pairs"] = {"shape": "mapping records -> fill_cigar -> clipIllegal / Boundary / OverlappingPairs -> CIGAR + MD:Z, "
                                       "m=150 flank=14, one call, nothing leaves the device", "pairs": np_pairs,
                              "total_ms": ms, "reads_per_s": na / ms * 1e3, "ok_fraction": float(fres["ok"].mean()),
                              "perfect_fraction": float((fres["method"] < 2).mean()),
                              "roofline": roof_hbm(na, 64 + 48 + q(150) + q(180) + float(fres["ext_cigar_len"].mean()) +
                                                   float(fres["cigar_len"].mean() + fres["mdz_len"].mean()), ms,
                                                   "finish_prepare + align + indel_ext_cigar + finish_merge + clip_* + sam_cigar + finish_final")}
    # R1: mapped reads/s of the device leg (BASELINE.json's second metric), one chained pass over the same reads in
    # the reference's order: 8 suffix-array candidates per read (its locus and 7 others; SURVEY 3.4 implies 4 - 8)
    # through dynamicProgramming, the loop's bookkeeping of mapping_core_sa.cpp:241-344 -> the bytes of the bin's
    # result group (results.cpp:546-598), then every pair through store_mapped_pair_reads' finish to CIGAR + MD:Z.
    # What the host keeps (SA search, pairing, MAPQ, SAM text) is not in it.
    c_per = 8
    nt = na * c_per
    rng = np.random.default_rng(91)
    rt = np.zeros(nt, api.DIST_TASK)
    rt["read_idx"] = np.repeat(np.arange(na, dtype=np.uint32), c_per)
    loci = rng.integers(1 << 16, G - (1 << 16), nt)
    loci[0::c_per] = p0 - 12
    rt["ref_pos"] = loci.astype(np.uint32)
    rt["max_distance_in_ref"] = 150 + 2 * 12
    rt["max_ed"] = 12
    rt["flags"] = np.repeat(is_rc.astype(np.uint8), c_per)
    d_rt = dev_bytes(rt)
    d_rr = torch.zeros(nt * 2, dtype=torch.int32, device=device)
    ctx.dynamic_programming_dev(reads_fin, d_rt, nt, d_rr, last_symbol=-1)
    ctx.synchronize()
    ed = d_rr.view(nt, 2).cpu().numpy()
    cm = np.zeros(nt, api.COMMIT_TASK)   # what the host loop writes down per candidate once its verifier has answered
    cm["lev_pos"] = (loci + 12).astype(np.uint32)
    cm["lev_errors"] = ed[:, 0].astype(np.uint32)
    cm["max_mismatches"] = 12
    cm["pattern_len"] = 150
    cm["direct"] = np.repeat(~is_rc, c_per)
    d_cm = dev_bytes(cm)
    d_committed = torch.zeros(nt * api.MAPPING.itemsize, dtype=torch.uint8, device=device)
    d_rf = torch.arange(0, nt + 1, c_per, dtype=torch.int32, device=device)
    d_ids = torch.arange(na, dtype=torch.int64, device=device) + (7 << 32)
    gcap = na * 16 + nt * 11 + 64
    d_gout = torch.zeros(gcap, dtype=torch.uint8, device=device)
    d_gnb = torch.zeros(1, dtype=torch.int32, device=device)
    d_goff = torch.zeros(na + 1, dtype=torch.int32, device=device)
    d_gws = torch.zeros(ctx.result_groups_workspace_bytes(nt, na), dtype=torch.uint8, device=device)
    gp = api.GroupParams(5, 4, 8, 0)

    def verify_all():
        ctx.dynamic_programming_dev(reads_fin, d_rt, nt, d_rr, last_symbol=-1)

    def groups():
        ctx.commit_candidates_dev(nt, d_cm, None, d_committed)
        ctx.emit_result_groups_dev(nt, d_cm, d_committed, na, d_rf, d_ids, gp, d_gout, gcap, d_gnb, d_goff, d_gws)

    def finish():
        ctx.finish_pairs_dev(reads_fin, d_ft, np_pairs, d_fr, d_fe, 352, d_fc, 352, d_fm, 352, d_fw[padw:], fwb, params=prm)

    def chain():
        verify_all()
        groups()
        finish()

    ms_v = timed(verify_all, tag="R1_verify", units=nt)
    ms_g = timed(groups, tag="K_result_groups", units=nt)
    if gather_cabi:
        # --gather cabi: the bin's result-group bytes - what rank 0 of a sharded stage needs - through the library's own
        # RCCL gather as records of one byte each (one rank here: ncclCommInitRank, the all-gather of counts, the packed
        # copy at the root; N > 1 replaces the copy by ncclSend / ncclRecv)
        comm1 = api.Comm(ctx, 0, 1, api.comm_unique_id())
        d_gg = torch.zeros(gcap, dtype=torch.uint8, device=device)
        holder = {}
        ms_gg = timed(lambda: holder.__setitem__("c", comm1.gather_records_dev(d_gout, d_gnb, gcap, d_gg, record_bytes=1)))
        nbytes = int(holder["c"][0])
        out["K_result_groups_gather"] = {"shape": "wamd_gather_records_dev of the bin's result-group bytes (record_bytes = 1), one rank",
                                         "bytes": nbytes, "kernel_ms": ms_gg,
                                         "same_bytes": bool(torch.equal(d_gg[:nbytes], d_gout[:nbytes])),
                                         "roofline": roof_hbm(nbytes, 2, ms_gg, "device-to-device copy at the root")}
        comm1.destroy()
        del d_gg
    ms = timed(chain, tag="R1_mapped_reads", units=na)
    # the same chain with the verifier in accepted-only mode (banded kernels): the commit step only asks
    # cur_match <= max_mismatches (mapping_core_sa.cpp:241), so the group bytes are the same
    ctx.set_distance_mode(api.DISTANCE_ACCEPTED_ONLY)
    ms_va = timed(verify_all, tag="R1_verify_accepted_only", units=nt)
    ms_a = timed(chain, tag="R1_mapped_reads_accepted_only", units=na)
    ctx.synchronize()
    ed_a = d_rr.view(nt, 2).cpu().numpy()
    same_accepts = bool(((ed_a[:, 0] <= 12) == (ed[:, 0] <= 12)).all() and (ed_a[ed[:, 0] <= 12] == ed[ed[:, 0] <= 12]).all())
    ctx.set_distance_mode(api.DISTANCE_EXACT)
    out["K_result_groups"] = {"shape": "commit_candidates + result group of the bin (heap of 8 per read, 5-byte ids)",
                              "tasks": nt, "reads": na, "kernel_ms": ms_g, "candidates_per_s": nt / ms_g * 1e3,
                              "group_bytes": int(d_gnb.item()),
                              "roofline": roof_hbm(nt, 16 + 2 * 40 + 4 + int(d_gnb.item()) / nt, ms_g,
                                                   "commit_kernel + group_order_kernel + group_write_kernel")}
    out["R1_mapped_reads"] = {"shape": "150 bp pairs, %d candidates per read: dynamicProgramming (exact mode, k = 12) -> "
                                       "commit + result groups -> finish_pairs (fill_cigar, clipping, CIGAR + MD:Z); "
                                       "device leg only, inputs resident" % c_per,
                              "reads": na, "candidates_per_read": c_per, "total_ms": ms,
                              "mapped_reads_per_s": na / ms * 1e3,
                              "stages_ms": {"verify": ms_v, "result_groups": ms_g, "finish_pairs": out["N1_finish_pairs"]["total_ms"]},
                              "accepted_fraction": float((ed[:, 0] <= 12).mean()),
                              "accepted_only": {"total_ms": ms_a, "mapped_reads_per_s": na / ms_a * 1e3, "verify_ms": ms_va,
                                                "same_accepted_candidates": same_accepts}}
    reads_fin.free()
    del d_ft, d_fr, d_fe, d_fc, d_fm, d_fw, d_rt, d_rr, d_cm, d_committed, d_rf, d_ids, d_gout, d_gnb, d_goff, d_gws

    # S3 / S6: mate rescue with traceback - long windows, k' = 2 * floor(0.09 m) (W = mu + 7 sigma + 2k')
    def rescue(tag, m, k, W, seed, count):
        rd_ = synth.make_reads(codes, count, m, k, seed, true_frac=1.0, chunk=1 << 15)
        reads_ = ctx.upload_reads(*host(rd_["codes"], 1), api.READS_ASCII, orientations=3)
        t_ = synth.distance_tasks(rd_, m, k, W=W)
        off_ = np.random.default_rng(seed).integers(k, W - m - k, count)      # the mate sits anywhere in the window
        t_["ref_pos"] = (rd_["p"].cpu().numpy() - off_).astype(np.uint32)
        d_t_ = dev_bytes(t_)
        stride_ = 4 * ((2 * (m + k) + 8 + 3) // 4)
        d_r_ = torch.zeros(count * api.ALIGN_RESULT.itemsize, dtype=torch.uint8, device=device)
        d_c_ = torch.zeros(count * stride_, dtype=torch.uint8, device=device)
        wsb_ = ctx.align_workspace_bytes(reads_, count, max_ed=k)
        d_w_ = torch.zeros(wsb_, dtype=torch.uint8, device=device)
        ms_ = timed(lambda: ctx.align_dev(reads_, d_t_, count, d_r_, d_c_, stride_, d_w_, wsb_, last_symbol=0, refine=1),
                    tag=tag, units=count)
        res_ = d_r_.cpu().numpy().view(api.ALIGN_RESULT)
        cols_ = W + 1 - (m - 1) // 64
        dw_ = (m - 1) // 32 + 1
        out[tag] = {"shape": "mate rescue: dynamicProgramming + getExtCigar m=%d k=%d W=%d" % (m, k, W), "tasks": count,
                    "total_ms": ms_, "alignments_per_s": count / ms_ * 1e3, "ok_fraction": float(res_["ok"].mean()),
                    "mean_edit_distance": float(res_["edit_distance"][res_["ok"] == 1].mean()),
                    "roofline": roof_valu(count, cols_ * (dw_ * 10 + 20), ms_,
                                          "myers_distance_kernel<%d> over the whole window, then the traceback "
                                          "(its work is not in the model)" % dw_)}
        if tag == "S3_rescue_150bp":
            # the same hits through the one call find_pairs<LevMyers> becomes (sam_matching.cpp:379-457): window
            # arithmetic, align, Translate's verdict, the num_events tests, the mate's record, the two clipping passes
            thr_ = W - 2 * k
            rt_ = np.zeros(count, api.RESCUE_TASK)
            direct_ = t_["flags"] == 0
            start_ = t_["ref_pos"].astype(np.int64)
            rt_["mate_read_idx"] = t_["read_idx"]
            rt_["hit_pos"] = np.where(direct_, start_ - m + thr_, start_ - 1 + k).astype(np.uint32)
            rt_["distance_threshold"] = thr_
            rt_["seq_lo"] = 0
            rt_["seq_hi"] = genome_symbols
            rt_["seq_delta"] = 1
            rt_["ref_size"] = genome_symbols
            rt_["max_mate_edit_distance"] = k
            rt_["hit_num_events"] = 2
            rt_["paired_dir_direct"] = direct_
            d_rt_ = dev_bytes(rt_)
            d_rr_ = torch.zeros(count * api.RESCUE_RESULT.itemsize, dtype=torch.uint8, device=device)
            wsr_ = ctx.rescue_workspace_bytes(reads_, count, k)
            d_wr_ = torch.zeros(wsr_, dtype=torch.uint8, device=device)
            ms_c = timed(lambda: ctx.rescue_mates_dev(reads_, d_rt_, count, d_rr_, d_c_, stride_, d_wr_, wsr_),
                         tag="N2_rescue_chain", units=count)
            rr_ = d_rr_.cpu().numpy().view(api.RESCUE_RESULT)
            out["N2_rescue_chain"] = {"shape": "wamd_rescue_mates_dev on S3's hits: find_pairs<LevMyers>'s chain per hit "
                                               "(window, align in accepted-only mode, Translate, num_events, record, "
                                               "clipIllegal / clipBoundary)", "tasks": count, "total_ms": ms_c,
                                      "hits_per_s": count / ms_c * 1e3, "aligned_fraction": float(rr_["aligned"].mean()),
                                      "pair_candidate_fraction": float(rr_["ok"].mean()),
                                      "same_alignments_as_S3": bool(((rr_["aligned"] == 1) == (res_["ok"] == 1)).all())}
            del d_rt_, d_rr_, d_wr_
        reads_.free()

    rescue("S3_rescue_150bp", 150, 26, 660, 83, na // 4)
    rescue("S6_rescue_250bp", 250, 44, 700, 84, na // 4)

    # S7: the other two variants - LevMyers128 (m = 100) and LevMyers64 (m = 300), distance only
    for tag, m, k in (("S7_distance_100bp_M128", 100, 10), ("S7_distance_300bp_M64", 300, 20)):
        rd = synth.make_reads(codes, na, m, k, 85 + m, chunk=1 << 16)
        reads = ctx.upload_reads(*host(rd["codes"], 0), api.READS_NIBBLE, orientations=3)
        t = synth.distance_tasks(rd, m, k)
        d_t = dev_bytes(t)
        d_r = torch.zeros(na * 2, dtype=torch.int32, device=device)
        ms = timed(lambda: ctx.dynamic_programming_dev(reads, d_t, na, d_r, last_symbol=-1), tag=tag, units=na)
        cols_ = (m + 2 * k + 1 - (m - 1) // 64) if m < 256 else m + 2 * k
        dw_ = (m - 1) // 32 + 1
        out[tag] = {"shape": "distance m=%d k=%d W=%d" % (m, k, m + 2 * k), "tasks": na, "kernel_ms": ms,
                    "candidates_per_s": na / ms * 1e3,
                    "roofline": roof_valu(na, cols_ * (dw_ * 10 + 20), ms, "myers_distance_kernel<%d>" % dw_)}
        reads.free()
        del d_t, d_r

    # S8: CIndelMatching::Match, max_indel = 50
    rc, fields = synth.indel_match_workload(codes, na, 150, 80)
    reads = ctx.upload_reads(*host(rc, 0), api.READS_NIBBLE, orientations=1)
    t = np.zeros(na, api.INDEL_TASK)
    for k_, v in fields.items():
        t[k_] = v
    d_t = dev_bytes(t)
    d_r = torch.zeros(na * api.MAPPING.itemsize, dtype=torch.uint8, device=device)
    prm_i = api.default_indel_params()
    prm_i.max_indel_bound = 50
    ms = timed(lambda: ctx.indel_match_dev(reads, d_t, na, d_r, params=prm_i), tag="S8_indel_match", units=na)
    res = d_r.cpu().numpy().view(api.MAPPING)
    out["S8_indel_match"] = {"shape": "CIndelMatching::Match m=150 max_indel=50 k=12", "tasks": na, "kernel_ms": ms,
                             "candidates_per_s": na / ms * 1e3, "found_fraction": float((res["found"] != 0).mean()),
                             "roofline": roof_hbm(na, q(150) + q(250) + 16 + 40, ms, "indel_match_kernel<true>")}
    # K: candidate_mapping_t::serialize of what Match found (results.cpp:161-217)
    d_dir = torch.ones(na, dtype=torch.uint8, device=device)
    d_wire = torch.zeros(na * 12, dtype=torch.uint8, device=device)
    d_sz = torch.zeros(na, dtype=torch.uint8, device=device)
    ms = timed(lambda: ctx.serialize_mappings_dev(na, d_r, d_dir, d_wire, d_sz), tag="K_serialize", units=na)
    out["K_serialize"] = {"shape": "candidate_mapping_t::serialize of S8's results", "records": na, "kernel_ms": ms,
                          "records_per_s": na / ms * 1e3, "roofline": roof_hbm(na, 40 + 1 + 12 + 1, ms, "serialize_kernel")}
    # G: GetExtCigarNew of the found mappings (fill_cigar_with_indel_matching, sam_matching.cpp:824-852)
    idx = np.nonzero(res["found"])[0]
    ct = np.zeros(len(idx), api.INDEL_CIGAR_TASK)
    ct["read_idx"] = t["read_idx"][idx]
    ct["examined_start_pos"] = res["pos"][idx]
    for f in ("type", "matching_len1", "matching_len2", "indel_len1", "indel_len2", "clipping_len"):
        ct[f] = res[f][idx]
    ng = len(idx)
    d_ct = dev_bytes(ct)
    d_cr = torch.zeros(ng * api.INDEL_CIGAR_RESULT.itemsize, dtype=torch.uint8, device=device)
    d_cc = torch.zeros(ng * 512, dtype=torch.uint8, device=device)
    ms = timed(lambda: ctx.indel_ext_cigar_dev(reads, d_ct, ng, d_cr, d_cc, 512), tag="G_indel_ext_cigar", units=ng)
    cres = d_cr.cpu().numpy().view(api.INDEL_CIGAR_RESULT)
    out["G_indel_ext_cigar"] = {"shape": "GetExtCigarNew of S8's found mappings", "records": ng, "kernel_ms": ms,
                                "records_per_s": ng / ms * 1e3, "ok_fraction": float(cres["ok"].mean()),
                                "roofline": roof_hbm(ng, 40 + q(150) + q(200) + 32 + float(cres["cigar_len"].mean()), ms,
                                                     "indel_ext_cigar_kernel")}
    reads.free()
    del d_t, d_r, d_dir, d_wire, d_sz, d_ct, d_cr, d_cc

    # row S: the hashing mate rescue, m = 150, W = 660 (mu + 7 sigma + 2k' of a 400 +- 30 library)
    ns = na // 4
    rc, fields = synth.softclip_workload(codes, ns, 150, 660, 81)
    reads = ctx.upload_reads(*host(rc, 1), api.READS_ASCII, orientations=1)
    t = np.zeros(ns, api.SOFTCLIP_TASK)
    for k_, v in fields.items():
        t[k_] = v
    d_t = dev_bytes(t)
    d_r = torch.zeros(ns * api.SOFTCLIP_RESULT.itemsize, dtype=torch.uint8, device=device)
    d_c = torch.zeros(ns * 384, dtype=torch.uint8, device=device)
    ms = timed(lambda: ctx.softclip_match_dev(reads, d_t, ns, d_r, d_c, 384, 660), tag="S_softclip_match", units=ns)
    res = d_r.cpu().numpy().view(api.SOFTCLIP_RESULT)
    out["S_softclip_match"] = {"shape": "CSoftClipping::match + getExtCigar m=150 W=660", "tasks": ns, "kernel_ms": ms,
                               "candidates_per_s": ns / ms * 1e3, "found_fraction": float((res["found"] == 1).mean()),
                               "indel_fraction": float((res["del_size"] != 0).mean()),
                               "roofline": roof_hbm(ns, q(150) + q(660 + 150) + 16 + 48 + 150, ms, "softclip_match_kernel")}
    reads.free()
    # per-kernel HBM traffic / VALU instructions from the committed PMC profile of this tree, where there is one
    meas = measured_counters()
    for tag, e in out.items():
        m_ = meas.get(tag)
        if m_ and "roofline" in e:
            units = e.get("tasks") or e.get("records") or e.get("pairs")
            if "hbm_bytes_per_unit" in m_:
                e["roofline"]["traffic"] = m_["hbm_bytes_per_unit"] * units
            if "valu_insts_per_unit" in m_:   # issued VALU lane-ops (wave instructions x 64) against the VALU peak
                e["roofline"]["measured_valu_frac"] = (m_["valu_insts_per_unit"] * units /
                                                       (e.get("kernel_ms", e.get("total_ms")) * 1e-3) / PEAK_VALU_LANEOPS)
            for k_ in ("sq_wait_any_frac", "sq_wait_inst_any_frac", "sq_active_inst_any_frac"):
                if k_ in m_:
                    e["roofline"][k_] = m_[k_]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tasks", type=int, default=1 << 24, help="candidates per GPU and step")
    ap.add_argument("--genome-symbols", type=int, default=1 << 28)
    ap.add_argument("--cpu-sample", type=int, default=1 << 24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary shapes (S2, S4, S5, S8)")
    ap.add_argument("--extras-tasks", type=int, default=1 << 22)
    ap.add_argument("--distance-mode", choices=("exact", "accepted-only"), default="exact",
                    help="what the timed step reports for rejected candidates (experiments / profiling; "
                         "the headline is measured in the default, exact mode)")
    ap.add_argument("--true-frac", type=float, default=0.5,
                    help="fraction of candidates planted at their locus (S1 = 0.5; other values are experiments)")
    ap.add_argument("--no-boundary", action="store_true", help="skip the host-buffer (PCIe-inclusive) measurements")
    ap.add_argument("--profile-markers", action="store_true",
                    help="bracket every timed region with marker kernels (for scripts/summarize_extras.py; profiling only)")
    ap.add_argument("--gather", choices=("torch", "cabi"), default="torch",
                    help="N > 1: the record gather through torch.distributed (RCCL behind it; pipelined behind the next "
                         "bin) or through the library's own C-ABI (wamd_gather_records_dev over librccl, what the "
                         "C++ host calls)")
    args = ap.parse_args()
    MARKS["on"] = args.profile_markers

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a CHILD process, before
    # anything here has touched the GPU, and relay the one JSON line rank 0 prints
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        child = subprocess.run(cmd, stdout=subprocess.PIPE)
        sys.stdout.write(child.stdout.decode())
        sys.stdout.flush()
        raise SystemExit(child.returncode)

    # stdout carries the one JSON line and nothing else: libraries that write to file descriptor 1
    # (RCCL prints its version banner there at the first collective) are sent to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = env_int("RANK", 0)
    world = env_int("WORLD_SIZE", 1)
    local = env_int("LOCAL_RANK", 0)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the verification path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    # WAMD_BENCH_FORCE_GATHER=1 rehearses the multi-rank code path (process group, record
    # emission, gather) with whatever world size the launcher gives, also 1
    multi = world > 1 or os.environ.get("WAMD_BENCH_FORCE_GATHER") == "1"
    if multi:
        import torch.distributed as dist_mod
        dist = dist_mod
        for k_, v_ in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k_, v_)   # (the one-rank rehearsal without a launcher)
        dist.init_process_group("nccl", device_id=device)

    n = args.tasks
    packed, ref_size, blob, off, lens, tasks = build_workload(device, n, args.genome_symbols,
                                                              seed=1234 + rank, true_frac=args.true_frac)
    ctx = api.Context(packed, ref_size=ref_size, cur_offset=0, device=local)   # genome borrowed on device
    ctx.set_max_text_len(M + 2 * K_ERR + 2)   # mapping_core.cpp:158-160
    if args.distance_mode != "exact":
        ctx.set_distance_mode(api.DISTANCE_ACCEPTED_ONLY)
    reads = ctx.upload_reads(blob, off, lens, api.READS_NIBBLE, orientations=3)
    d_tasks = torch.from_numpy(tasks.view(np.uint8)).to(device)
    d_results = torch.zeros(n * 2, dtype=torch.int32, device=device)
    d_results_alt = None   # second result buffer of the multi-rank path (see exchange_step)
    stream = torch.cuda.ExternalStream(ctx.stream(), device=device)
    torch.cuda.synchronize()

    def verify(out=None):
        ctx.dynamic_programming_dev(reads, d_tasks, n, d_results if out is None else out, last_symbol=-1)

    # mapping records of the accepted candidates (results.cpp:161-217) -> rank 0: the path's
    # only exchange step.  Positions / strands are what the host loop would record
    # (mapping_core_sa.cpp:243, 298); here: the window start + k on the direct strand.
    d_rec_pos = torch.from_numpy((tasks["ref_pos"] + K_ERR).astype(np.uint32).view(np.int32)).to(device)
    d_dir = torch.ones(n, dtype=torch.uint8, device=device)
    n_buf = 2 if multi else 1
    d_records = [torch.zeros((n, gather.RECORD_BYTES), dtype=torch.uint8, device=device) for _ in range(n_buf)]
    d_count = [torch.zeros(1, dtype=torch.int32, device=device) for _ in range(n_buf)]
    d_scratch = torch.zeros(ctx.records_scratch_bytes(n), dtype=torch.uint8, device=device)

    # The exchange of bin i runs behind the verification of bin i + 1: bins alternate between two
    # result buffers, the records of bin i are emitted on a second stream as soon as its
    # verification is through, the host only looks at their count once bin i + 1 has been
    # enqueued, and the gather itself - straight out of the record buffer, two of those in turn
    # as well - stays in flight until its buffer is needed again.  The last one is waited for
    # before the clock stops.
    if multi:
        d_results_alt = torch.zeros(n * 2, dtype=torch.int32, device=device)
    res_bufs = [d_results, d_results_alt]
    side_e = torch.cuda.Stream(device=device) if multi else None   # record emission
    side_g = torch.cuda.Stream(device=device) if multi else None   # counts and gather
    verified = [torch.cuda.Event() for _ in range(n_buf)]
    emitted = [torch.cuda.Event() for _ in range(n_buf)]
    pend = [None] * n_buf
    bins_done = [0]

    comm = d_gathered = None
    if multi and args.gather == "cabi":
        ids = [api.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        comm = api.Comm(ctx, rank, world, ids[0])
        d_gathered = torch.zeros((n * world if rank == 0 else 1, gather.RECORD_BYTES), dtype=torch.uint8, device=device)

    class _Done:
        def result(self):
            return None

    def start_gather(b):
        with torch.cuda.stream(side_g):
            side_g.wait_event(emitted[b])
            if comm is not None:      # waits for the counts (one read-back), then enqueues the sends / receives
                comm.gather_records_dev(d_records[b], d_count[b], n, d_gathered, root=0, stream=side_g.cuda_stream)
                side_g.synchronize()
                pend[b] = _Done()
            else:
                pend[b] = gather.gather_records(d_records[b], d_count[b], dst=0, async_op=True, copy=False)

    def exchange_step():
        k = bins_done[0]
        b = k & 1
        if k >= 2:
            stream.wait_event(emitted[b])   # bin k - 2 has been read out of this result buffer
        verify(res_bufs[b])
        verified[b].record(stream)
        if k > 0:
            start_gather((k - 1) & 1)       # the host reads the counts of bin k - 1 while bin k runs
        with torch.cuda.stream(side_e):
            side_e.wait_event(verified[b])
            if pend[b] is not None:         # the gather of bin k - 2 still reads this record buffer
                pend[b].result()
                pend[b] = None
            ctx.emit_lev_records_dev(n, d_tasks, res_bufs[b], d_rec_pos, d_dir, d_records[b], d_count[b], d_scratch,
                                     stream=side_e.cuda_stream)
            emitted[b].record(side_e)
        bins_done[0] = k + 1

    def finish_gathers():
        if multi and bins_done[0] > 0:
            start_gather((bins_done[0] - 1) & 1)
            with torch.cuda.stream(stream):
                for b in range(n_buf):
                    if pend[b] is not None:
                        pend[b].result()
                        pend[b] = None
            bins_done[0] = 0

    def step():
        if multi:
            exchange_step()
        else:
            verify()

    for _ in range(args.warmup):
        step()
    finish_gathers()
    ctx.synchronize()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()

    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(args.steps)]
    main_tag = "S1_exact" if args.distance_mode == "exact" else "S1_accepted_only"
    mark(main_tag, True, stream, device, args.steps, n)   # (no-ops unless --profile-markers; never in a judged run)
    gc.collect()
    gc.disable()   # no collection pause inside the timed steps
    t0 = time.perf_counter()
    for a, b in evs:
        # (in the multi-rank path the bracket also holds the wait for the buffer's last reader)
        a.record(stream)
        step()
        b.record(stream)
    mark(main_tag, False, stream, device)
    last_buf = res_bufs[(args.steps - 1) & 1] if multi else d_results
    finish_gathers()
    ctx.synchronize()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    if dist:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    flags = ctx.take_flags()
    if flags:
        raise SystemExit("device error flags 0x%x" % flags)

    res = last_buf.view(n, 2).cpu().numpy()
    accepted = int((res[:, 0] <= K_ERR).sum())

    # the same step in accepted-only mode (banded kernels): accepted candidates must come back
    # identical, the others as {k + 1, 0}
    banded = None
    if rank == 0 and world == 1 and not args.no_extras and args.distance_mode == "exact":
        ctx.set_distance_mode(api.DISTANCE_ACCEPTED_ONLY)
        verify()
        ctx.synchronize()
        bev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        mark("S1_accepted_only", True, stream, device, 5, n)
        for a, b in bev:
            a.record(stream)
            verify()
            b.record(stream)
        mark("S1_accepted_only", False, stream, device)
        ctx.synchronize()
        ctx.set_distance_mode(api.DISTANCE_EXACT)
        bres = d_results.view(n, 2).cpu().numpy()
        rej = res[:, 0] > K_ERR
        same = bool((bres[~rej] == res[~rej]).all() and (bres[rej, 0] == K_ERR + 1).all() and (bres[rej, 1] == 0).all())
        bms = float(np.mean([a.elapsed_time(b) for a, b in bev]))
        banded = {"shape": "S1 in WAMD_DISTANCE_ACCEPTED_ONLY mode: 64-row band, %d sliding + %d fixed columns"
                           % (BAND_FS, N_COLS - BAND_FS),
                  "tasks": n, "kernel_ms": bms, "candidates_per_s": n / bms * 1e3,
                  "accepted_identical_rejected_clamped": same,
                  "roofline": roof_valu(n, LANE_OPS_BAND, bms, "myers_band_kernel<2,256,true>")}
        banded["roofline"]["note"] = ("the model counts every candidate's whole band; since the end of round 3 candidates the "
                                      "band shows hopeless stop at the test column (96 of 173) and their block's survivors are "
                                      "packed, so at this workload's 50 % related candidates `frac` overstates the issued work "
                                      "by up to a quarter - the time is what is measured")
        banded["roofline"]["algorithmic"] = {"lane_ops_per_task": LANE_OPS_BAND_RECURRENCE,
                                             "frac": n * LANE_OPS_BAND_RECURRENCE / (bms * 1e-3) / PEAK_VALU_LANEOPS}
        banded["roofline"]["hbm"] = {"achieved": n * BYTES_PER_TASK / (bms * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9,
                                     "unit": "GB/s", "frac": n * BYTES_PER_TASK / (bms * 1e-3) / PEAK_HBM}

    if rank == 0:
        value = n * world * args.steps / dt
        out = {
            "metric": "verified candidates/sec",
            "value": value,
            "unit": "candidates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": "S1: LevMyers256 distance verification, 150 bp reads, k = 12, W = 174 "
                            "(sensitive stages of BASELINE configs[1]); %d candidates per GPU per step, "
                            "%d-symbol random genome with N runs, %d%% true candidates with 0..15 edits, "
                            "50%% reverse strand" % (n, args.genome_symbols, round(100 * args.true_frac)),
                "candidates_per_gpu": n,
                "distance_mode": args.distance_mode,
                "accepted_fraction": accepted / n,
                "parallelism": "bins sharded over %d GPU(s), no data-path collective; "
                               "accepted records gathered to rank 0 over RCCL (%s)" % (world, args.gather),
            },
            "gcups": value * M * N_COLS / 1e9,
        }
        exact = args.distance_mode == "exact"
        meas = measured_counters().get("S1_exact" if exact else "S1_accepted_only", {})
        out["roofline"] = roof_valu(n, LANE_OPS_EXACT if exact else LANE_OPS_BAND, kernel_ms,
                                    "myers_distance_kernel<%d,256>" % DWORDS if exact else "myers_band_kernel<2,256,true>")
        out["roofline"]["kernel_ms"] = kernel_ms
        alg = LANE_OPS_EXACT_RECURRENCE if exact else LANE_OPS_BAND_RECURRENCE
        out["roofline"]["algorithmic"] = {"lane_ops_per_task": alg,
                                          "achieved": n * alg / (kernel_ms * 1e-3) / 1e12,
                                          "frac": n * alg / (kernel_ms * 1e-3) / PEAK_VALU_LANEOPS,
                                          "note": "recurrence only (10 lane-ops per 32-bit dword and column; band: 9 "
                                                  "sliding / 10 fixed), no bookkeeping: distance from an algorithmic "
                                                  "bound, where `frac` above is issue-slot utilisation"}
        out["roofline"]["survey_8d"] = {"lane_ops_per_task": LANE_OPS_SURVEY_8D,
                                        "frac": n * LANE_OPS_SURVEY_8D / (kernel_ms * 1e-3) / PEAK_VALU_LANEOPS,
                                        "note": "SURVEY 8(d)'s count for the same task; not a lower bound on gfx950"}
        out["roofline"]["hbm"] = {"achieved": n * BYTES_PER_TASK / (kernel_ms * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9,
                                  "unit": "GB/s", "frac": n * BYTES_PER_TASK / (kernel_ms * 1e-3) / PEAK_HBM,
                                  "algorithmic_bytes_per_task": BYTES_PER_TASK}
        if "hbm_bytes_per_unit" in meas:
            out["roofline"]["traffic"] = meas["hbm_bytes_per_unit"] * n
        if "valu_insts_per_unit" in meas:
            out["roofline"]["measured_valu_frac"] = meas["valu_insts_per_unit"] * n / (kernel_ms * 1e-3) / PEAK_VALU_LANEOPS
        if not args.no_cpu_baseline and world == 1:   # (the contract: rank 0 at N = 1 only)
            threads = max(1, min(16, os.cpu_count() or 1))
            base, cpu_res = cpu_baseline(packed.cpu().numpy(), ref_size, blob, off, lens, tasks,
                                         min(args.cpu_sample, n), threads)
            # the checker also checks: the GPU results of the sample equal the CPU's
            k = len(cpu_res)
            ce, cp = cpu_res["edit_distance"], cpu_res["pos"]
            ge, gp = res[:k, 0].astype(np.uint32), res[:k, 1].astype(np.uint32)
            if exact:
                same = bool((ce == ge).all() and (cp == gp).all())
            else:   # accepted-only mode: what the reference accepts comes back identical, the rest as {k + 1, 0}
                acc_ = ce <= K_ERR
                same = bool((ce[acc_] == ge[acc_]).all() and (cp[acc_] == gp[acc_]).all() and
                            (ge[~acc_] == K_ERR + 1).all() and (gp[~acc_] == 0).all())
            base["gpu_matches_cpu_on_sample"] = same
            out["cpu_baseline"] = base
            if not same:
                out["parity_error"] = "GPU and CPU results differ on the baseline sample"
        if not args.no_boundary and world == 1:
            out["boundary"] = boundary_costs(ctx, device, blob, off, lens, tasks, n)
            out["boundary"]["compact"] = boundary_compact(ctx, device, blob, off, lens, tasks, n, ref_size)
        if not args.no_extras and world == 1:
            del d_tasks, d_results, d_results_alt, res_bufs, last_buf, d_records, d_scratch, d_rec_pos, d_dir, d_count
            reads.free()
            reads = None
            torch.cuda.empty_cache()
            out["extras"] = run_extras(ctx, device, args.genome_symbols, stream, args.extras_tasks, gather_cabi=args.gather == "cabi")
            out["extras"]["S1_accepted_only"] = banded
            r1 = out["extras"].get("R1_mapped_reads")
            if r1 and "cpu_baseline" in out:
                # the reference's verifier alone on the host cores, the same candidates per read: an upper bound
                # on what the CPU leg can map (its traceback and post-passes come on top)
                r1["cpu_verify_only_reads_per_s"] = out["cpu_baseline"]["value"] / r1["candidates_per_read"]
                r1["cpu_cores"] = out["cpu_baseline"]["cores"]
        if MARKS["on"]:
            out["profile_markers"] = {"tags": MARKS["tags"], "reps": MARKS["reps"], "units": MARKS["units"]}
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
        if out.get("parity_error"):
            raise SystemExit(1)
    if comm is not None:
        comm.destroy()
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if reads is not None:
        reads.free()
    ctx.destroy()


if __name__ == "__main__":
    main()
