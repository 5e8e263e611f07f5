"""Print the few figures of a bench.py JSON line that a kernel experiment looks at (headline + every other config)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]


def show(name, ms, kern, first, roof, plan):
    print("%s: %.4f ms/step  kernel_ms %s  first count %s  plan %.1f ms  frac %.3f (traffic %s)" % (
        name, ms, kern, first, plan, roof["frac"], roof.get("traffic")))


show(c["workload"][:2], d["ms_per_step"], c["kernel_ms"], c.get("first_count_ms"), d["roofline"], c["plan_build_ms_once_per_annotation"])
if "two_files" in c:
    print("  two files:", c["two_files"])
for k, o in c.get("other_configs", {}).items():
    show(k, o["ms_per_step"], o["kernel_ms"], o.get("first_count_ms"), o["roofline"], o["plan_build_ms_once_per_annotation"])
print("  scopes:", {k: (round(v) if isinstance(v, (int, float)) else v) for k, v in c["scopes"].items() if not isinstance(v, dict)})
