"""The S4 workload of bench.py alone (fill_cigar_with_lev: distance + traceback + ext-CIGAR + refinement, m = 150):
for profiler passes.   usage: python scripts/run_align.py [tasks] [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from whisper_amd import api, synth  # noqa: E402

na = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
codes = synth.make_genome(1 << 28, seed=0xC0FFEE, device=dev)
packed = synth.pack_nibbles(codes)
ctx = api.Context(packed, ref_size=codes.numel(), cur_offset=0, device=0)
ctx.set_max_text_len(16384)
stream = torch.cuda.ExternalStream(ctx.stream(), device=dev)
rd = synth.make_reads(codes, na, 150, 10, 79, true_frac=1.0, chunk=1 << 16)
blob, off, lens = synth.pack_reads(rd["codes"], 1)
reads = ctx.upload_reads(blob.cpu().numpy(), off.cpu().numpy().astype(np.uint64), lens.cpu().numpy().astype(np.uint32),
                         api.READS_ASCII, orientations=3)
t = synth.distance_tasks(rd, 150, 13, W=177)
t["ref_pos"] = (rd["p"] - 13).cpu().numpy().astype(np.uint32)
d_t = torch.from_numpy(np.ascontiguousarray(t).view(np.uint8).reshape(-1)).to(dev)
stride = 512
d_r = torch.zeros(na * api.ALIGN_RESULT.itemsize, dtype=torch.uint8, device=dev)
d_c = torch.zeros(na * stride, dtype=torch.uint8, device=dev)
wsb = ctx.align_workspace_bytes(reads, na)
d_w = torch.zeros(wsb, dtype=torch.uint8, device=dev)
ctx.align_dev(reads, d_t, na, d_r, d_c, stride, d_w, wsb, last_symbol=0, refine=1)
ctx.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for a, b in evs:
    a.record(stream)
    ctx.align_dev(reads, d_t, na, d_r, d_c, stride, d_w, wsb, last_symbol=0, refine=1)
    b.record(stream)
ctx.synchronize()
res = d_r.cpu().numpy().view(api.ALIGN_RESULT)
print("S4 n=%d ms=%.4f ok=%.4f ws=%.1f B/task flags=0x%x checksum=%d cigsum=%d" % (
    na, float(np.mean([a.elapsed_time(b) for a, b in evs])), float(res["ok"].mean()), wsb / na, ctx.take_flags(),
    int(res["begin_pos"].astype(np.uint64).sum() * 3 + res["cigar_len"].astype(np.uint64).sum() * 7 + res["num_events"].sum() * 131
        + res["refined"].sum() * 1009), int(d_c.sum(dtype=torch.int64))))
