#!/usr/bin/env python3
"""Pretty-print the interesting parts of a bench.py JSON line."""
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"):
        continue
    j = json.loads(line)
    print("value %.2f %s  n_gpus %d  ms/step %.2f" % (j["value"], j["unit"], j["n_gpus"], j["ms_per_step"]))
    print("roofline", j["roofline"])
    print("pipeline", j["extra"].get("roofline_pipeline"))
    print("cpu", j["cpu_baseline"])
    print("bf", j["extra"]["bf_match"])
    for k, v in j["extra"]["kernels"].items():
        print("   %-20s %s" % (k, v))
    if j["extra"].get("bow"):
        print("bow", j["extra"]["bow"])
    ba = j["extra"].get("ba")
    if ba:
        print("ba", {k: v for k, v in ba.items() if k != "kernels"})
        for k, v in ba["kernels"].items():
            print("   %-20s %s" % (k, v))
