"""Debug aid: one case of scripts/fuzz_parity.py's saloop family, compact form with per-read k against the task form."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import fuzz_parity as fz
from oracle import bindings as ob
from whisper_amd import api
import test_gpu_sa_loop as T

label, z = fz.saloop_case(np.random.default_rng(int(sys.argv[1])), ob.Reference())
print(label)
k, seg, mrl = int(z["k"]), (int(z["seg"][0]), int(z["seg"][1])), int(z["min_read_len"])
n, nr = len(z["commit"]), len(z["read_ids"])
lev = k < 7
counts, sa = T._compact(z)
oracle = ob.Oracle()
for filtered in (True, False):
    cf = z["cf"].copy()
    if not filtered:
        cf["pattern_len"] = 0
    ctx = api.Context(z["packed"], ref_size=int(z["ref_size"]), cur_offset=seg[0])
    reads = ctx.upload_reads(z["blob"], z["off"], z["lens"], api.READS_NIBBLE, orientations=3)
    b = api.sa_batch(nr, k, seg, mrl if filtered else 0, api.SA_VERIFIER_LEVDIAG if lev else api.SA_VERIFIER_DISTANCE,
                     reverse_read_offset=nr)
    rng = np.random.default_rng(3)
    read_k = rng.integers(max(1, k - 4) if lev else max(7, k - 5), k + 1, nr).astype(np.uint8)
    per_cand = np.repeat(read_k, counts.sum(axis=1).astype(np.int64))
    dist_t = z["dist"].copy(); dist_t["max_ed"] = per_cand
    lev_t = z["lev"].copy(); lev_t["max_mismatches"] = per_cand
    got2 = ctx.verify_sa_candidates(reads, b, counts, sa, read_k=read_k)
    res = {}
    for mode in (api.DISTANCE_EXACT, api.DISTANCE_ACCEPTED_ONLY):
        ctx.set_distance_mode(mode)
        wsb = ctx.filter_verify_workspace_bytes(n)
        d_ws = torch.zeros(wsb + 256, dtype=torch.uint8, device="cuda:0"); d_ws = d_ws[(-d_ws.data_ptr()) % 256:]
        d_cf, d_dist2 = T.dev(cf), T.dev(dist_t)
        d_res = torch.zeros(n * 2, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        ts = torch.cuda.Stream()
        with torch.cuda.stream(ts):
            ctx.filter_verify_distance_dev(reads, d_cf, d_dist2, n, d_res, d_ws, wsb, last_symbol=-1, stream=ts.cuda_stream)
        torch.cuda.synchronize()
        res[mode] = d_res.cpu().numpy().view(np.uint32).reshape(n, 2).copy()
        # plain distance without the filter
        r2 = ctx.dynamic_programming(reads, dist_t, -1)
        res[(mode, "plain")] = r2
    rc, want = oracle.distance(z["packed"], z["blob"], z["off"], z["lens"], 0, dist_t, -1)
    e2 = res[api.DISTANCE_EXACT][:, 0]
    want2 = np.where(e2 <= per_cand, e2, per_cand.astype(np.uint32) + 1).astype(np.uint8)
    bad = np.nonzero(got2 != want2)[0]
    print("filtered", filtered, "bad", bad)
    for i in bad:
        print(" cand", i, "task", dist_t[i], "m", z["lens"][dist_t[i]["read_idx"]], "cf", cf[i], "compact", got2[i], "task exact", res[api.DISTANCE_EXACT][i],
              "task acc", res[api.DISTANCE_ACCEPTED_ONLY][i], "plain exact", res[(api.DISTANCE_EXACT, "plain")][i],
              "plain acc", res[(api.DISTANCE_ACCEPTED_ONLY, "plain")][i], "oracle", want[i])
    reads.free(); ctx.destroy()
