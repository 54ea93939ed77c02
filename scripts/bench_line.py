"""Condense one bench.py JSON line (stdin) to a short summary (used when sweeping kernel variants)."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
j = json.loads(sys.stdin.read())
r = j["roofline"]
print(tag, round(j["value"], 1), "pivots/s; K", j["config"].get("pivots_per_sweep"), "sweep",
      round(r["avg_kernel_ms"] * 1e3, 1), "us x", r["launches_sampled"], "obj", j["objective_after_timed_region"])
