import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import common
from whisper_amd import api
from oracle import bindings as ob
m,k,W,last,refine,variant = [int(x) if x!='None' else None for x in sys.argv[1:7]]
n = int(sys.argv[7]) if len(sys.argv)>7 else 4000
codes, packed, (blob, off, lens), tasks = common.distance_workload(m, k, n, seed=300+m+k, encoding=1, W=W, variant=variant)
stride = 4 * ((2 * (m + k) + 8 + 3) // 4)
O = ob.Oracle()
rc, want, wcig = O.align(packed, blob, off, lens, 1, tasks, last, refine, cigar_stride=stride)
ctx = api.Context(packed, ref_size=codes.numel())
reads = ctx.upload_reads(blob, off, lens, 1)
got, gcig = ctx.align(reads, tasks, last, refine, cigar_stride=stride)
bad = [i for i in range(n) if got[i] != want[i] or bytes(gcig[i]).split(b'\0')[0] != bytes(wcig[i]).split(b'\0')[0]]
print('bad', len(bad))
for i in bad[:6]:
    print(i, tasks[i]); print(' got ', got[i], bytes(gcig[i]).split(b'\0')[0]); print(' want', want[i], bytes(wcig[i]).split(b'\0')[0])
