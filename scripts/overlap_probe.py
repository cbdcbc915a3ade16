"""Debug aid: do wamd_reads_refill (second thread) and wamd_verify_sa_candidates overlap?  Timestamps per bin."""
import sys, os, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_amd import api, synth
import bench
dev = torch.device("cuda:0")
n = 1 << 23
packed, ref_size, blob, off, lens, tasks = bench.build_workload(dev, n, 1 << 26, 7)
ctx = api.Context(packed, ref_size=ref_size, cur_offset=0, device=0)
c, k = 8, 12
fl, fr = synth.stage_segments(150, k)[1]
per = 1 << 19
bins = 2
n_reads = per * bins
rng = np.random.default_rng(1)
sa = rng.integers(1 << 16, ref_size - (1 << 16), (n_reads, c)).astype(np.uint32)
counts = np.zeros((n_reads, 2), np.uint16); counts[:, 0] = c
P = api.PinnedBuffer
p_counts, p_sa, p_err = P(2 * n_reads, np.uint16), P(n_reads * c, np.uint32), P(n_reads * c, np.uint8)
p_counts.array[:] = counts.reshape(-1); p_sa.array[:] = sa.reshape(-1)
o1 = int(off[n_reads - 1]) + 75
p_blob, p_off, p_lens = P(o1 + 64, np.uint8), P(n_reads, np.uint64), P(n_reads, np.uint32)
p_blob.array[:o1] = blob[:o1]; p_off.array[:] = off[:n_reads]; p_lens.array[:] = lens[:n_reads]
offs = [np.ascontiguousarray(p_off.array[b * per:(b + 1) * per] - p_off.array[b * per]) for b in range(bins)]
up = torch.cuda.Stream()
batch = [ctx.upload_reads(p_blob.array[int(p_off.array[b * per]):], offs[b], p_lens.array[b * per:(b + 1) * per], 0, 3, stream=up.cuda_stream) for b in range(2)]
log = []
def refill(b):
    t0 = time.perf_counter()
    batch[b & 1].refill(p_blob.array[int(p_off.array[b * per]):], offs[b], p_lens.array[b * per:(b + 1) * per], stream=up.cuda_stream)
    log.append(("refill", b, t0, time.perf_counter()))
def verify(b):
    t0 = time.perf_counter()
    a, e = b * per, (b + 1) * per
    ctx.verify_sa_candidates(batch[b & 1], api.sa_batch(per, k, (fl, fr), 0, 0), p_counts.array[2 * a:2 * e], p_sa.array[a * c:e * c], errors=p_err.array[a * c:e * c])
    log.append(("verify", b, t0, time.perf_counter()))
verify(0); verify(1)
for rep in range(3):
    log.clear()
    T0 = time.perf_counter()
    refill(0)
    for b in range(4):
        th = threading.Thread(target=refill, args=((b + 1) & 1,)); th.start()
        verify(b & 1)
        th.join()
    T1 = time.perf_counter()
    for what, b, a, e in sorted(log, key=lambda x: x[2]):
        print(rep, what, b, "start %.3f ms  dur %.3f ms" % ((a - T0) * 1e3, (e - a) * 1e3))
    print(rep, "all %.3f ms" % ((T1 - T0) * 1e3))
t0 = time.perf_counter(); refill(1); print("refill alone %.3f" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); verify(0); print("verify alone %.3f" % ((time.perf_counter() - t0) * 1e3))
