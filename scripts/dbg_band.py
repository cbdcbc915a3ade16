"""Debug: N-dense genome, exact vs accepted-only mode against the oracle."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import common
from whisper_amd import api, synth
from oracle import bindings
orc = bindings.Oracle()
codes = synth.make_genome(1 << 18, seed=9, n_frac=0.05, n_run=7)
packed = synth.pack_nibbles(codes).numpy()
c = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
for m, k, enc in ((150, 12, 0), (150, 12, 1), (250, 17, 0), (100, 10, 0), (300, 12, 0)):
    rd = synth.make_reads(codes, 4000, m, k, seed=40 + m, n_read_frac=0.2)
    blob, off, lens = common.host_reads(rd, enc)
    tasks = synth.distance_tasks(rd, m, k)
    rc, want = orc.distance(packed, blob, off, lens, enc, tasks, -1)
    reads = c.upload_reads(blob, off, lens, enc)
    for mode in (0, 1):
        c.set_distance_mode(mode)
        got = c.dynamic_programming(reads, tasks, -1)
        w = want.copy()
        if mode:
            rej = w["edit_distance"] > k
            w["edit_distance"][rej] = k + 1
            w["pos"][rej] = 0
        bad = np.flatnonzero((got["edit_distance"] != w["edit_distance"]) | (got["pos"] != w["pos"]))
        print(m, k, enc, "mode", mode, "bad", len(bad), [(int(i), tuple(got[i]), tuple(w[i])) for i in bad[:4]])
        for i in bad[:3]:
            t = tasks[i]
            win = codes[int(t["ref_pos"]):int(t["ref_pos"]) + int(t["max_distance_in_ref"]) + 2].numpy()
            print("   window tail", "".join("ACGTNnXX"[x & 7] for x in win[-30:]), "read tail",
                  "".join("ACGTNnXX"[x & 7] for x in rd["codes"][i].numpy()[-12:]), "rc", int(t["flags"]))
    reads.free()
