"""Diagnostic: one line per bench JSON line on stdin (value, time per step, IK kernel time, host enqueue time)."""
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith("=="):
        print(l)
    elif l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print("  value %.3e  ms/step %.4f  ik_ms %.4f frac %.3f enq_us %.2f" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], d["config"].get("host_enqueue_us_per_step", -1)))
