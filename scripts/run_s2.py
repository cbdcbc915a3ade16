"""The S2 workload of bench.py alone (LevDiag, m = 150, k = 5): for profiler passes on that kernel.
usage: python scripts/run_s2.py [tasks] [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from whisper_amd import api, synth  # noqa: E402

if os.environ.get("WAMD_LIB"):
    api.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", os.environ["WAMD_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
codes = synth.make_genome(1 << int(os.environ.get("WAMD_S2_GENOME_LOG2", "28")), seed=0xC0FFEE, device=dev)
packed = synth.pack_nibbles(codes)
ctx = api.Context(packed, ref_size=codes.numel(), cur_offset=0, device=0)
stream = torch.cuda.ExternalStream(ctx.stream(), device=dev)
fl, fr = synth.stage_segments(150, 5)[2]
rd = synth.make_reads(codes, n, 150, 5, 77, rc_frac=0.0, protect=(fl, fr), chunk=1 << 17, true_frac=float(os.environ.get("WAMD_TRUE_FRAC", "0.5")))
blob, off, lens = synth.pack_reads(rd["codes"], 0)
reads = ctx.upload_reads(blob.cpu().numpy(), off.cpu().numpy().astype(np.uint64), lens.cpu().numpy().astype(np.uint32),
                         api.READS_NIBBLE, orientations=1)
t = synth.levdiag_tasks(rd, 5, fl, fr)
d_t = torch.from_numpy(np.ascontiguousarray(t).view(np.uint8).reshape(-1)).to(dev)
d_r = torch.zeros(n, dtype=torch.int32, device=dev)
ctx.lev_diag_dev(reads, d_t, n, d_r)
ctx.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for a, b in evs:
    a.record(stream)
    ctx.lev_diag_dev(reads, d_t, n, d_r)
    b.record(stream)
ctx.synchronize()
r = d_r.cpu().numpy().view(np.uint32)
print("S2 n=%d ms=%.4f accepted=%.4f flags=0x%x checksum=%d" % (
    n, float(np.mean([a.elapsed_time(b) for a, b in evs])), float((r <= 5).mean()), ctx.take_flags(), int(r.astype(np.uint64).sum())))
