"""Debug aid: times wamd_softclip_match_dev on three pure workloads (unrelated / junk prefix / deletion)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from whisper_amd import api, synth  # noqa: E402

dev = torch.device("cuda", 0)
n = 1 << 18
codes = synth.make_genome(1 << 26, seed=0xC0FFEE, device=dev)
packed = synth.pack_nibbles(codes)
ctx = api.Context(packed.cpu().numpy(), ref_size=codes.numel(), cur_offset=0)
stream = torch.cuda.ExternalStream(ctx.stream())
rc, fields = synth.softclip_workload(codes, 3 * n, 150, 660, 81)
for kind, name in ((2, "unrelated"), (0, "junk prefix"), (1, "deletion")):
    sel = torch.arange(kind, 3 * n, 3, device=dev)
    blob, off, lens = synth.pack_reads(rc[sel], 1)
    reads = ctx.upload_reads(blob.cpu().numpy(), off.cpu().numpy().astype(np.uint64), lens.cpu().numpy().astype(np.uint32),
                             api.READS_ASCII, orientations=1)
    t = np.zeros(n, api.SOFTCLIP_TASK)
    for k_, v in fields.items():
        t[k_] = v[kind::3]
    t["read_idx"] = np.arange(n, dtype=np.uint32)
    d_t = torch.from_numpy(t.view(np.uint8)).to(dev)
    d_r = torch.zeros(n * api.SOFTCLIP_RESULT.itemsize, dtype=torch.uint8, device=dev)
    d_c = torch.zeros(n * 384, dtype=torch.uint8, device=dev)
    ms = []
    for i in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        ctx.softclip_match_dev(reads, d_t, n, d_r, d_c, 384, 660)
        b.record(stream)
        ctx.synchronize()
        ms.append(a.elapsed_time(b))
    res = d_r.cpu().numpy().view(api.SOFTCLIP_RESULT)
    print("%-12s %.3f ms per %d tasks, found %.2f" % (name, min(ms), n, (res["found"] == 1).mean()))
    reads.free()
