"""Debug aid: wamd_reads_refill of one bin (2^19 reads of 150 symbols), wall time per call and, under
rocprofv3 --kernel-trace --memory-copy-trace --stats, where it goes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from whisper_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
codes = synth.make_genome(1 << 24, seed=1, device=dev)
packed = synth.pack_nibbles(codes)
ctx = api.Context(packed, ref_size=codes.numel(), cur_offset=0, device=0)
rd = synth.make_reads(codes, n, 150, 12, 7)
blob, off, lens = synth.pack_reads(rd["codes"], 0)
blob, off, lens = blob.cpu().numpy(), off.cpu().numpy().astype(np.uint64), lens.cpu().numpy().astype(np.uint32)
pb, po, pl = api.PinnedBuffer(len(blob) + 64, np.uint8), api.PinnedBuffer(n, np.uint64), api.PinnedBuffer(n, np.uint32)
pb.array[:len(blob)] = blob
po.array[:] = off[:n]
pl.array[:] = lens
st = torch.cuda.Stream(device=dev)
reads = ctx.upload_reads(pb.array, po.array, pl.array, api.READS_NIBBLE, orientations=3, stream=st.cuda_stream)
for _ in range(3):
    reads.refill(pb.array, po.array, pl.array, stream=st.cuda_stream)
t0 = time.perf_counter()
for _ in range(reps):
    reads.refill(pb.array, po.array, pl.array, stream=st.cuda_stream)
dt = (time.perf_counter() - t0) / reps
print("refill of %d reads: %.3f ms per call, %d bytes of reads (%.1f GB/s if it were all copy)" % (n, dt * 1e3, len(blob), (len(blob) + 12 * n) / dt / 1e9))
# the copy alone, for scale
d = torch.empty(len(blob), dtype=torch.uint8, device=dev)
h = torch.from_numpy(pb.array[:len(blob)])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
print("torch copy of the same bytes: %.3f ms" % ((time.perf_counter() - t0) / reps * 1e3), "pinned:", h.is_pinned())
reads.free()
ctx.destroy()
