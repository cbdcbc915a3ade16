"""Debug aid: times S1 (and S5) in both distance modes with the library named in WAMD_LIB
(default: the product library).  Run on the GPU box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from whisper_amd import api, synth  # noqa: E402

if os.environ.get("WAMD_LIB"):
    api.LIB_PATH = os.path.join(ROOT, os.environ["WAMD_LIB"])
tag = sys.argv[1] if len(sys.argv) > 1 else "product"
dev = torch.device("cuda", 0)
n = 1 << 23
codes = synth.make_genome(1 << 27, seed=0xC0FFEE, device=dev)
packed = synth.pack_nibbles(codes)
ctx = api.Context(packed, ref_size=codes.numel(), cur_offset=0, device=0)
stream = torch.cuda.ExternalStream(ctx.stream(), device=dev)
for m, k in ((150, 12), (250, 17), (150, 5)):
    rd = synth.make_reads(codes, n, m, k, 1234, chunk=1 << 16, true_frac=float(os.environ.get("WAMD_TRUE_FRAC", "0.5")))
    blob, off, lens = synth.pack_reads(rd["codes"], 0)
    reads = ctx.upload_reads(blob.cpu().numpy(), off.cpu().numpy().astype(np.uint64),
                             lens.cpu().numpy().astype(np.uint32), api.READS_NIBBLE, orientations=3)
    t = synth.distance_tasks(rd, m, k)
    d_t = torch.from_numpy(t.view(np.uint8)).to(dev)
    d_r = torch.zeros(n * 2, dtype=torch.int32, device=dev)
    out = []
    for mode in (0, 1):
        ctx.set_distance_mode(mode)
        ms = []
        for rep in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            ctx.dynamic_programming_dev(reads, d_t, n, d_r, last_symbol=-1)
            b.record(stream)
            ctx.synchronize()
            ms.append(a.elapsed_time(b))
        out.append(min(ms[1:]))
    print("%-12s m=%d k=%d  exact %.3f ms  accepted-only %.3f ms  (2^23 tasks)" % (tag, m, k, out[0], out[1]), flush=True)
    reads.free()
    del d_t, d_r
ctx.destroy()
