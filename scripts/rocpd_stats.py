#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace into the --stats style table (per-kernel calls,
total / average / min / max duration, % of GPU time).  Usage: rocpd_stats.py results.db [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(
        f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for name, calls, tot, avg, mn, mx in rows:
        lines.append(f'"{name}",{calls},{tot},{avg:.1f},{mn},{mx},{100.0 * tot / total:.2f}')
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    sys.stdout.write(out)


if __name__ == "__main__":
    main()
