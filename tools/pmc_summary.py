#!/usr/bin/env python3
"""Summarise a rocprofv3 results .db (kernel trace and/or PMC) into a compact text table.
usage: pmc_summary.py results.db [kernel-name-substring]"""
import sqlite3, sys
db = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
con = sqlite3.connect(db); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
if "top_kernels" in tabs:
    print("== kernel summary (name, calls, total_ns/us?, avg, pct) ==")
    for r in cur.execute("select * from top_kernels limit 12"):
        print("  %-70s calls=%-6s total=%-14s avg=%-12s pct=%.2f" % (str(r[0])[:70], r[1], r[2], r[3], r[4]))
pmc_tabs = [t for t in tabs if t.startswith("rocpd_pmc_event") or t == "counters_collection" or "pmc" in t.lower()]
print("pmc tables:", pmc_tabs[:6])
if "counters_collection" in tabs:
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    print(cols)
    q = "select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection %s group by kernel_name, counter_name" % (
        ("where kernel_name like '%%%s%%'" % pat) if pat else "")
    for r in cur.execute(q):
        print("  %-50s %-28s n=%-5d sum=%-18.6g avg=%.6g" % (str(r[0])[:50], r[1], r[2], r[3], r[4]))
