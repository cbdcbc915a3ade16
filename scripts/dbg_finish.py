"""Debug aid: the finish-chain batch of a fuzz seed; finds the pairs on which wamd_finish_pairs raises an error the
oracle does not, by halving, and prints them."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import common
from oracle import bindings as ob
from whisper_amd import api

rng = np.random.default_rng(int(sys.argv[1]))
oracle = ob.Oracle()
m = int(rng.choice([100, 150, 151, 250]))
flank = int(rng.integers(4, 15))
fpf = float(rng.choice([0.0, 0.7]))
name = rng.choice(["default", "asym", "odd"])
sc_o, sc_a = common.scoring(name), common.scoring(name, api)
(codes, packed), (blob, off, lens), desc, ridx, maps, direct, distant, stale = common.finish_workload(
    oracle, 1000, int(rng.integers(1, 1 << 30)), m=m, flank=flank, scoring=sc_o, from_pair_frac=fpf)
kw = dict(isize=common.FINISH_ISIZE) if fpf else {}
tasks = common.finish_tasks(desc, ridx, lens, maps, direct, distant, stale, scoring=sc_o, **kw)
print("m", m, "flank", flank, "from_pair", fpf, "scoring", name, "pairs", len(tasks) // 2)
rc, *want = oracle.finish_pairs(packed, blob, off, lens, tasks, ob.default_finish_params(flank), scoring=sc_o, ext_stride=768)
print("oracle rc", rc)
ctx = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
ctx.set_scoring(sc_a or api.default_scoring())
reads = ctx.upload_reads(blob, off, lens, 1, orientations=3)

def fails(lo, hi):
    try:
        ctx.finish_pairs(reads, tasks[2 * lo:2 * hi], api.default_finish_params(flank), ext_stride=768)
        return False
    except api.WamdError:
        return True

todo, badp = [(0, len(tasks) // 2)], []
while todo:
    lo, hi = todo.pop()
    if not fails(lo, hi):
        continue
    if hi - lo == 1:
        badp.append(lo)
        continue
    mid = (lo + hi) // 2
    todo += [(lo, mid), (mid, hi)]
print("failing pairs", badp)
res, ext, cig, mdz = want
for p in badp[:4]:
    for sl in (2 * p, 2 * p + 1):
        print(" slot", sl, "task", tasks[sl])
        print("   oracle result", res[sl], bytes(ext[sl]).split(b"\0")[0], bytes(cig[sl]).split(b"\0")[0], bytes(mdz[sl]).split(b"\0")[0])
