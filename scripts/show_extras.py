"""Debug aid: prints the timings and rooflines of a bench line (default gpurun_out/b.json)."""
import json
import sys

d = json.loads(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/b.json").read().strip().splitlines()[-1])
print("S1 %.3f ms/step, %.3g candidates/s, roofline.frac %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["frac"]))
for k, v in d.get("extras", {}).items():
    ms = v.get("kernel_ms", v.get("total_ms"))
    units = v.get("tasks") or v.get("records") or v.get("pairs") or v.get("reads")
    r = v.get("roofline", {})
    print("%-26s %8.3f ms  %10.4g units/s  %-5s frac %.3f" % (k, ms, units / ms * 1e3, r.get("bound", "-"), r.get("frac", 0)))
for k, v in d.get("boundary", {}).items():
    print("boundary.%-45s %s" % (k, v))
