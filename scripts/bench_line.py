"""Condense one bench.py JSON line (stdin) to a short summary (used when sweeping kernel variants)."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
j = json.loads(sys.stdin.read())
r = j["roofline"]
par = j.get("parity_after_timed_region") or {}
print(tag, round(j["value"], 1), "pivots/s; K", j["config"].get("pivots_per_sweep"), "sweep",
      round((r.get("avg_kernel_ms") or 0) * 1e3, 1), "us x", r["launches_sampled"], "bound", r["bound"], "frac",
      None if r["frac"] is None else round(r["frac"], 3), "parity", par.get("ok", par.get("reason")),
      "engine", j.get("engine"))
if "cfg3" in j:
    r3 = j["cfg3"]["roofline"]
    print(tag, "cfg3", round(j["cfg3"]["value"], 1), "pivots/s sweep", round((r3.get("avg_kernel_ms") or 0) * 1e3, 1),
          "us frac", None if r3["frac"] is None else round(r3["frac"], 3), "parity",
          (j["cfg3"].get("parity_after_timed_region") or {}).get("ok"))
for group in ("steady", "steady_plain", "steady_fused"):
  for name, leg in (j.get(group) or {}).items():
    if not isinstance(leg, dict) or "roofline" not in leg:
        if isinstance(leg, dict):
            print(tag, group, name, leg)
        continue
    rs = leg["roofline"]
    print(tag, group, name, round(leg["value"], 1), "pivots/s;", rs.get("kernel"), round((rs.get("avg_kernel_ms") or 0) * 1e3, 1),
          "us x", rs["launches_sampled"], "frac", None if rs["frac"] is None else round(rs["frac"], 3), "bound", rs.get("bound"),
          "clock", rs.get("clock_ghz"), "Mcycles", None if not rs.get("cycles_per_launch") else round(rs["cycles_per_launch"] / 1e6, 3),
          "parity", (leg.get("parity_after_timed_region") or {}).get("ok"))
if isinstance(j.get("onepass"), dict):
    o = j["onepass"]
    print(tag, "onepass", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in o.items() if k in ("value", "steps", "error")},
          "parity", (o.get("parity_after_timed_region") or {}).get("ok"))
