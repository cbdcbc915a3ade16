"""Verification aid: wamd_softclip_match against the oracle on SC_STRESS_N windows (default 20 000) per shape - random and
low-complexity genomes, both object-history states, reads of 100 - 250 symbols, windows of 300 - 5 000.  Run on the GPU
box: gpurun -- python scripts/softclip_stress.py (2.1 M windows bit-equal at SC_STRESS_N=300000, round 3)."""
import sys, os, time
root = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, root + "/tests"); sys.path.insert(0, root)
import numpy as np
import common
from oracle import bindings as ob
from whisper_amd import api
orc = ob.Oracle()
bad = 0
for lowcx, fresh, W, m, mql, seed in ((1, 0, 700, 150, 151, 1), (0, 0, 660, 150, 151, 2), (1, 1, 1427, 150, 151, 3), (1, 0, 300, 100, 100, 4),
                                       (1, 0, 2000, 250, 251, 5), (0, 1, 900, 150, 163, 6), (1, 0, 5000, 150, 151, 7)):
    gen = common.low_complexity_genome() if lowcx else common.genome()
    t0 = time.time()
    codes, packed, (blob, off, lens), tasks = common.softclip_workload(int(os.environ.get("SC_STRESS_N", "20000")), seed=7000 + seed, m=m, W=W, encoding=1, gen=gen, lens_jitter=True)
    prm_o = ob.default_softclip_params(max_query_len=mql, fresh_object=fresh)
    _, want, wcig = orc.softclip_match(packed, blob, off, lens, 1, tasks, prm_o, cigar_stride=768)
    ctx = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    reads = ctx.upload_reads(blob, off, lens, 1, orientations=3)
    got, gcig = ctx.softclip_match(reads, tasks, api.default_softclip_params(max_query_len=mql, fresh_object=fresh), cigar_stride=768)
    reads.free(); ctx.destroy()
    same = all((got[f] == want[f]).all() for f in want.dtype.names)
    cig_same = all(bytes(gcig[i, :want["cigar_len"][i] + 1]) == bytes(wcig[i, :want["cigar_len"][i] + 1]) for i in range(len(want)))
    print("lowcx %d fresh %d W %d m %d: found %d, results equal %s, cigars equal %s (%.1f s)" % (lowcx, fresh, W, m, int(want["found"].sum()), same, cig_same, time.time() - t0), flush=True)
    bad += (not same) + (not cig_same)
sys.exit(1 if bad else 0)
