"""Kernel statistics (name, calls, total/avg/min/max ns, share) from a rocprofv3 rocpd .db -> CSV on stdout.
usage: python tools/rocpd_stats.py results.db [> profiles/xyz_kernel_stats.csv]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = con.execute(f"select {name}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                   f"from kernels group by {name} order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for n, c, t, a, lo, hi in rows:
    n = n.split("(")[0]
    print(f'"{n}",{c},{t},{a:.1f},{100.0 * t / total:.2f},{lo},{hi}')
