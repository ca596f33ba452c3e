#!/usr/bin/env python3
"""profiles/pmc_json.py <workload> <dir with pmc_sq.csv, pmc_sq_trace.csv, pmc_fetch.csv, pmc_write.csv, kernel_stats.csv>
-> JSON on stdout: per kernel the average counter values per launch, the average launch duration of the counter pass
(kernel trace of that same pass) and of the plain kernel-trace run, and the derived HBM bytes
(2 x FETCH_SIZE + WRITE_SIZE: gfx950 counts a wide streaming read at half its bytes, guides/MI355X_MICROARCH.md)."""
import csv
import json
import os
import subprocess
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def counters(path):
    agg = defaultdict(lambda: defaultdict(list))
    if not os.path.exists(path):
        return agg
    for r in csv.DictReader(open(path)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    wl, d = sys.argv[1], sys.argv[2]
    out = defaultdict(dict)
    for f in ("pmc_sq.csv", "pmc_fetch.csv", "pmc_write.csv", "pmc_mix.csv"):
        for k, cs in counters(os.path.join(d, f)).items():
            for c, v in cs.items():
                out[k][c] = round(sum(v) / len(v), 1)
                out[k].setdefault("launches", {})[c] = len(v)
    tr = os.path.join(d, "pmc_sq_trace.csv")
    if os.path.exists(tr):
        dur = defaultdict(list)
        for r in csv.DictReader(open(tr)):
            dur[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        for k, v in dur.items():
            out[k]["counter_pass_avg_ns"] = round(sum(v) / len(v), 1)
    ks = os.path.join(d, "kernel_stats.csv")
    if os.path.exists(ks):
        for r in csv.DictReader(open(ks)):
            out[short(r["Name"])]["kernel_trace_avg_ns"] = float(r["AverageNs"])
            out[short(r["Name"])]["kernel_trace_calls"] = int(r["Calls"])
    for k, v in out.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:  # both in KB
            v["hbm_bytes_per_launch"] = int((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
        if "GRBM_GUI_ACTIVE" in v and v.get("counter_pass_avg_ns"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; reads high on launches well under 0.3 ms (guide, DVFS section)
            v["clock_ghz_counter_pass"] = round(v["GRBM_GUI_ACTIVE"] / 8.0 / v["counter_pass_avg_ns"], 3)
    head = os.environ.get("SVGR_HEAD")  # (the GPU box has no .git: the caller names the commit)
    try:
        head = head or subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:  # noqa: BLE001  (the GPU box has no .git)
        head = None
    # what the counters were collected FROM: bench.py compares this with the tree it runs in (`counters.stale`)
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("svgr_hip.hip", "svgr_core.h"):
        with open(os.path.join(root, "svgrasterize.py_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    print(json.dumps({"workload": wl, "collected_by": "profiles/collect2.sh", "head": head, "source_sha256": h.hexdigest(),
                      "units": "FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them; SQ_* raw; *_ns nanoseconds",
                      "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
