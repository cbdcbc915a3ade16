"""Debug aid: accepted-only distance on small batches (one wave, one block, several) against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from oracle import bindings as ob
from whisper_amd import api
oracle = ob.Oracle()
codes, packed, (blob, off, lens), tasks = common.distance_workload(150, 12, 1024, seed=262, true_frac=float(sys.argv[1]) if len(sys.argv) > 1 else 0.5)
ctx = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
ctx.set_distance_mode(1)
reads = ctx.upload_reads(blob, off, lens, 0)
rc, want = oracle.distance(packed, blob, off, lens, 0, tasks, -1)
k = tasks["max_ed"]
rej = want["edit_distance"] > k
want["edit_distance"][rej] = k[rej] + 1
want["pos"][rej] = 0
for n in (64, 128, 256, 512, 1024):
    got = ctx.dynamic_programming(reads, tasks[:n], -1)
    bad = np.flatnonzero((got["edit_distance"] != want["edit_distance"][:n]) | (got["pos"] != want["pos"][:n]))
    acc = (~rej[:n])
    print("n", n, "accepted", int(acc.sum()), "per wave", [int(acc[i:i + 64].sum()) for i in range(0, n, 64)], "bad", len(bad), bad[:20].tolist())
    for i in bad[:4]:
        print("   task", i, "got", got[i], "want", want[i])
