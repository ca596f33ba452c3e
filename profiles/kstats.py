#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite output): profiles/kstats.py results.db [csv-out]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by name order by 6 desc").fetchall()
tot = sum(r[5] for r in rows)
lines = ["kernel,calls,avg_us,min_us,max_us,total_us,percent"]
for r in rows:
    lines.append(f"\"{r[0]}\",{r[1]},{r[2]/1e3:.2f},{r[3]/1e3:.2f},{r[4]/1e3:.2f},{r[5]/1e3:.1f},{100*r[5]/tot:.1f}")
out = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
for l in lines:
    print(l[:150])
