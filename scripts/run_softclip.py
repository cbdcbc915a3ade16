"""The row-S workload of bench.py alone (CSoftClipping::match + getExtCigar, m = 150, W = 660): for profiler passes.
usage: python scripts/run_softclip.py [windows] [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from whisper_amd import api, synth  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
codes = synth.make_genome(1 << 28, seed=0xC0FFEE, device=dev)
packed = synth.pack_nibbles(codes)
ctx = api.Context(packed, ref_size=codes.numel(), cur_offset=0, device=0)
ctx.set_max_text_len(16384)
stream = torch.cuda.ExternalStream(ctx.stream(), device=dev)
rc, fields = synth.softclip_workload(codes, ns, 150, 660, 81)
blob, off, lens = synth.pack_reads(rc, 1)
reads = ctx.upload_reads(blob.cpu().numpy(), off.cpu().numpy().astype(np.uint64), lens.cpu().numpy().astype(np.uint32),
                         api.READS_ASCII, orientations=1)
t = np.zeros(ns, api.SOFTCLIP_TASK)
for k_, v in fields.items():
    t[k_] = v
d_t = torch.from_numpy(t.view(np.uint8).reshape(-1)).to(dev)
d_r = torch.zeros(ns * api.SOFTCLIP_RESULT.itemsize, dtype=torch.uint8, device=dev)
d_c = torch.zeros(ns * 384, dtype=torch.uint8, device=dev)
ctx.softclip_match_dev(reads, d_t, ns, d_r, d_c, 384, 660)
ctx.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for a, b in evs:
    a.record(stream)
    ctx.softclip_match_dev(reads, d_t, ns, d_r, d_c, 384, 660)
    b.record(stream)
ctx.synchronize()
res = d_r.cpu().numpy().view(api.SOFTCLIP_RESULT)
chk = int(res["pos"].astype(np.uint64).sum() + res["cigar_len"].astype(np.uint64).sum() * 7 + res["no_mismatches"].sum() * 131)
print("S ns=%d ms=%.4f found=%.4f indel=%.4f flags=0x%x checksum=%d cigsum=%d" % (
    ns, float(np.mean([a.elapsed_time(b) for a, b in evs])), float((res["found"] == 1).mean()),
    float((res["del_size"] != 0).mean()), ctx.take_flags(), chk, int(d_c.sum(dtype=torch.int64))))
