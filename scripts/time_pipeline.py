"""Debug aid: the with-upload pipeline of bench.py's boundary_compact (async refill of bin b + 1 beside the compact
verification of bin b) on its own, for a copy / kernel timeline under rocprofv3 --kernel-trace --memory-copy-trace."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from whisper_amd import api, synth

order = sys.argv[1] if len(sys.argv) > 1 else "refill-first"
bins, per, c, k, M = 4, 1 << 19, 8, 12, 150
dev = torch.device("cuda", 0)
codes = synth.make_genome(1 << 26, seed=1, device=dev)
packed = synth.pack_nibbles(codes)
ref_size = codes.numel()
ctx = api.Context(packed, ref_size=ref_size, cur_offset=0, device=0)
ctx.set_max_text_len(M + 2 * k + 2)
n_reads = bins * per
rd = synth.make_reads(codes, n_reads, M, k, 7, rc_frac=0.0)
blob, off, lens = synth.pack_reads(rd["codes"], 0)
blob, off, lens = blob.cpu().numpy(), off.cpu().numpy().astype(np.uint64), lens.cpu().numpy().astype(np.uint32)
fl, fr = synth.stage_segments(M, k)[1]
rng = np.random.default_rng(1)
ref_pos = np.empty((n_reads, c), np.int64)
ref_pos[:, 0] = rd["p"].cpu().numpy() - k
ref_pos[:, 1:] = rng.integers(1 << 16, ref_size - (1 << 16), (n_reads, c - 1))
sa = (ref_pos + k + fl).astype(np.uint32)
counts = np.zeros((n_reads, 2), np.uint16)
counts[:, 0] = c
P = api.PinnedBuffer
p_counts, p_sa, p_err = P(2 * n_reads, np.uint16), P(n_reads * c, np.uint32), P(n_reads * c, np.uint8)
p_counts.array[:] = counts.reshape(-1)
p_sa.array[:] = sa.reshape(-1)
p_blob, p_off, p_lens = P(len(blob) + 64, np.uint8), P(n_reads, np.uint64), P(n_reads, np.uint32)
p_blob.array[:len(blob)] = blob
p_lens.array[:] = lens
stride = int(off[1])
for b in range(bins):
    p_off.array[b * per:(b + 1) * per] = off[b * per:(b + 1) * per] - off[b * per]
up = torch.cuda.Stream(device=dev)
batch = [ctx.upload_reads(p_blob.array[b * per * stride:], p_off.array[b * per:(b + 1) * per], p_lens.array[b * per:(b + 1) * per],
                          api.READS_NIBBLE, orientations=3, stream=up.cuda_stream) for b in (0, 1)]

def refill(b):
    a, e = b * per, (b + 1) * per
    batch[b & 1].refill_async(p_blob.array[a * stride:], per * stride, p_off.array[a:e], p_lens.array[a:e], M, stream=up.cuda_stream)

def verify(b):
    a, e = b * per, (b + 1) * per
    ctx.verify_sa_candidates(batch[b & 1], api.sa_batch(per, k, (fl, fr), 0, api.SA_VERIFIER_DISTANCE),
                             p_counts.array[2 * a:2 * e], p_sa.array[a * c:e * c], errors=p_err.array[a * c:e * c])

def pipeline():
    refill(0)
    up.synchronize()
    for b in range(bins):
        if order == "refill-first" and b + 1 < bins:
            refill(b + 1)
        verify(b)
        if order == "none":
            continue
        up.synchronize()

for rep in range(3):
    t0 = time.perf_counter()
    pipeline()
    dt = time.perf_counter() - t0
    print(order, "%.3f ms per bin, %.2f G candidates/s" % (dt / bins * 1e3, bins * per * c / dt / 1e9), "accepted", float((p_err.array <= k).mean()))
