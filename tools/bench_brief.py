import json, sys
d = json.loads(sys.stdin.read())
r = d["roofline"]
print(sys.argv[1] if len(sys.argv) > 1 else "", "ms/step", d["ms_per_step"], "kernel", r.get("kernel"), "kernel_ms", r.get("kernel_ms_avg"),
      "frac", r["frac"], "Mpix/s", d["value"])
