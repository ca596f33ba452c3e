"""Which planner the batches of a document take, and whether the two-pass plan's add guess holds (SVGR_DBG_PLAN=1 output counted)."""
import os, sys, subprocess, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for wl in ("material4096", "icons4096", "tiger2048"):
    env = dict(os.environ, SVGR_DBG_PLAN="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", "2", "--warmup", "0", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    c = collections.Counter()
    for ln in r.stderr.splitlines():
        if ln.startswith("[plan] two passes: capacity bits"):
            c["two-pass ok" if " bits 0 " in ln else "two-pass FELL BACK"] += 1
            if " bits 0 " not in ln: print("   ", ln[:220])
        elif ln.startswith("[plan] single pass"):
            c["single pass ok" if "err 0 " in ln else "single pass flagged"] += 1
    print(wl, dict(c))
