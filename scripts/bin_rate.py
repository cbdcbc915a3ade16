"""Host-to-host rate of the per-bin form through the C++ mirror (whisper_amd/host/replay_batch --bin): a synthetic bin of
2^19 reads of 150 symbols with 8 suffix-array entries each (each read's own locus first, both strands), k = 12, no CF
pre-filter (every candidate reaches the verifier, as in bench.py's S1).  Run through gpurun; prints replay_batch's lines."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from whisper_amd import synth  # noqa: E402

n_reads, c, m, k = 1 << 19, 8, 150, 12
dev = torch.device("cuda:0")
codes = synth.make_genome(1 << 26, seed=0xC0FFEE, device=dev)
packed = synth.pack_nibbles(codes).cpu().numpy()
ref_size = codes.numel()
rd = synth.make_reads(codes, n_reads, m, k, 5, true_frac=0.5, chunk=1 << 17)
blob, off, lens = synth.pack_reads(rd["codes"], 0)
blob, lens = blob.cpu().numpy(), lens.cpu().numpy().astype(np.uint32)
tasks = synth.distance_tasks(rd, m, k)
fl, fr = synth.stage_segments(m, k)[1]
rng = np.random.default_rng(1)
rc = (tasks["flags"] & 1).astype(bool)
ref_pos = np.empty((n_reads, c), np.int64)
ref_pos[:, 0] = tasks["ref_pos"]
ref_pos[:, 1:] = rng.integers(1 << 16, ref_size - (1 << 16), (n_reads, c - 1))
sa = np.where(rc[:, None], ref_size - m + fl - k - ref_pos, ref_pos + k + fl).astype(np.uint32)
counts = np.zeros((n_reads, 2), np.uint16)
counts[:, 0] = np.where(rc, 0, c)
counts[:, 1] = np.where(rc, c, 0)
stride = (m + 1) // 2
reads = blob[: n_reads * stride].reshape(n_reads, stride)


def vec(a):
    a = np.ascontiguousarray(a)
    return struct.pack("<Q", a.size) + a.tobytes()


with tempfile.TemporaryDirectory() as d:
    fin, fout = os.path.join(d, "bin.in"), os.path.join(d, "bin.out")
    with open(fin, "wb") as f:
        f.write(struct.pack("<8I", ref_size, fl, n_reads, fl, fr, 0, 7, stride))
        for a in (packed, reads, lens, np.full(n_reads, k, np.uint8), counts, sa.reshape(-1)):
            f.write(vec(a))
    del codes, rd
    torch.cuda.empty_cache()
    out = subprocess.run([os.path.join(ROOT, "whisper_amd", "host", "replay_batch"), "--bin", fin, fout, "4"],
                         stdout=subprocess.PIPE, text=True, check=True, timeout=300)
    print(out.stdout)
    err = np.fromfile(fout, np.uint8)
    print("accepted fraction", float((err <= k).mean()))
