#!/usr/bin/env python3
"""Dump the per-kernel summary of a rocprofv3 rocpd database (bench_results.db) as CSV: name,calls,total_us,avg_us,pct."""
import sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
with open(out, "w") as f:
    f.write("kernel,calls,total_us,avg_us,percent\n")
    for n, k, t, a, p in rows:
        f.write('"%s",%d,%.3f,%.3f,%.2f\n' % (n, k, t, a, p))
print(open(out).read())
