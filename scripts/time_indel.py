"""Debug aid: times wamd_indel_match_dev on shape S8 with the library named by $WAMD_LIB."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from whisper_amd import api, synth  # noqa: E402

if os.environ.get("WAMD_LIB"):
    api.LIB_PATH = os.path.join(ROOT, os.environ["WAMD_LIB"])
dev = torch.device("cuda", 0)
n = 1 << 20
codes = synth.make_genome(1 << 26, seed=0xC0FFEE, device=dev)
packed = synth.pack_nibbles(codes)
ctx = api.Context(packed, ref_size=codes.numel(), cur_offset=0, device=0)
rc, fields = synth.indel_match_workload(codes, n, 150, 80)
blob, off, lens = synth.pack_reads(rc, 0)
reads = ctx.upload_reads(blob.cpu().numpy(), off.cpu().numpy().astype(np.uint64), lens.cpu().numpy().astype(np.uint32),
                         api.READS_NIBBLE, orientations=1)
t = np.zeros(n, api.INDEL_TASK)
for k_, v in fields.items():
    t[k_] = v
d_t = torch.from_numpy(t.view(np.uint8)).to(dev)
d_r = torch.zeros(n * api.MAPPING.itemsize, dtype=torch.uint8, device=dev)
stream = torch.cuda.ExternalStream(ctx.stream())
ms = []
for i in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    ctx.indel_match_dev(reads, d_t, n, d_r)
    b.record(stream)
    ctx.synchronize()
    ms.append(a.elapsed_time(b))
res = d_r.cpu().numpy().view(api.MAPPING)
print("%-10s indel match ms: %s  found %.3f flags %x" % (sys.argv[1] if len(sys.argv) > 1 else "", " ".join("%.3f" % x for x in ms[1:]),
                                                      float((res["found"] != 0).mean()), ctx.take_flags()))
