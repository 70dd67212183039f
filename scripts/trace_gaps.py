#!/usr/bin/env python3
"""Idle time between consecutive shade launches of the headline loop, from a rocprofv3 kernel trace: python scripts/trace_gaps.py <kernel_trace.csv>"""
import csv, sys, statistics as st
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
sh = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_forward_lighting" in r["Kernel_Name"]))
gaps = [b[0] - a[1] for a, b in zip(sh, sh[1:])]
tail = gaps[-120:]
print("shade launches", len(sh), "median duration us", st.median(e - s for s, e in sh[-120:]) / 1e3)
print("gap between shade kernels (last 120): median %.1f us, p10 %.1f, p90 %.1f" % (st.median(tail) / 1e3, sorted(tail)[12] / 1e3, sorted(tail)[108] / 1e3))
pc = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_post_chain" in r["Kernel_Name"]))
print("post chain launches", len(pc), "median us", st.median(e - s for s, e in pc[-120:]) / 1e3)
