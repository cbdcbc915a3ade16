"""Debug aid: prints the secondary-shape timings of the bench line in gpurun_out/b.json."""
import json
import sys

d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/b.json"))
print("S1 %.3f ms/step, %.3g candidates/s, roofline.frac %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["frac"]))
for k, v in d.get("extras", {}).items():
    ms = v.get("kernel_ms", v.get("total_ms"))
    rate = v.get("candidates_per_s", v.get("alignments_per_s"))
    print("%-20s %8.3f ms  %.4g /s  (%d tasks)" % (k, ms, rate, v["tasks"]))
