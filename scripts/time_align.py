"""Debug aid: times wamd_align_dev on shape S4 with the library named by $WAMD_LIB."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from whisper_amd import api, synth  # noqa: E402

if os.environ.get("WAMD_LIB"):
    api.LIB_PATH = os.path.abspath(os.environ["WAMD_LIB"])
dev = torch.device("cuda", 0)
n = 1 << 20
codes = synth.make_genome(1 << 26, seed=0xC0FFEE, device=dev)
packed = synth.pack_nibbles(codes)
ctx = api.Context(packed.cpu().numpy(), ref_size=codes.numel(), cur_offset=0)
rd = synth.make_reads(codes, n, 150, 10, 79, true_frac=1.0, chunk=1 << 16)
blob, off, lens = synth.pack_reads(rd["codes"], 1)
reads = ctx.upload_reads(blob.cpu().numpy(), off.cpu().numpy().astype(np.uint64), lens.cpu().numpy().astype(np.uint32),
                         api.READS_ASCII, orientations=3)
t = synth.distance_tasks(rd, 150, 13, W=177)
t["ref_pos"] = (rd["p"] - 13).cpu().numpy().astype(np.uint32)
d_t = torch.from_numpy(t.view(np.uint8)).to(dev)
d_r = torch.zeros(n * api.ALIGN_RESULT.itemsize, dtype=torch.uint8, device=dev)
d_c = torch.zeros(n * 512, dtype=torch.uint8, device=dev)
wsb = ctx.align_workspace_bytes(reads, n)
d_w = torch.zeros(wsb, dtype=torch.uint8, device=dev)
stream = torch.cuda.ExternalStream(ctx.stream())
ms = []
for i in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    ctx.align_dev(reads, d_t, n, d_r, d_c, 512, d_w, wsb, last_symbol=0, refine=1)
    b.record(stream)
    ctx.synchronize()
    ms.append(a.elapsed_time(b))
print("%-16s align total ms:" % (sys.argv[1] if len(sys.argv) > 1 else ""), " ".join("%.3f" % x for x in ms[1:]))
