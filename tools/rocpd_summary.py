#!/usr/bin/env python
"""Summarise rocprofv3 rocpd (.db) output: per-kernel time stats and per-kernel PMC averages.

    python tools/rocpd_summary.py <results.db> [more.db ...]
"""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    name = name.replace("schpf::", "").replace("void ", "")
    return name[:70]


def main():
    for db in sys.argv[1:]:
        con = sqlite3.connect(db)
        cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
        print("##", db)
        if "start" in cols and "end" in cols:
            rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), "
                               "max(end-start) from kernels group by name order by 3 desc").fetchall()
            tot = sum(r[2] for r in rows) or 1
            print("%-72s %7s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us",
                                                         "max_us", "pct"))
            for n, c, s, a, mn, mx in rows:
                print("%-72s %7d %12.1f %12.2f %12.2f %12.2f %6.2f" % (short(n), c, s / 1e3, a / 1e3, mn / 1e3,
                                                                      mx / 1e3, 100.0 * s / tot))
        try:
            ccols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
            if ccols:
                q = ("select kernel_name, counter_name, count(*), avg(value), sum(value) from "
                     "counters_collection group by kernel_name, counter_name order by 1, 2")
                rows = con.execute(q).fetchall()
                if rows:
                    print("%-60s %-28s %7s %18s" % ("kernel", "counter", "n", "avg_per_dispatch"))
                for k, cn, n, a, s in rows:
                    print("%-60s %-28s %7d %18.1f" % (short(k)[:60], cn, n, a))
        except sqlite3.Error as e:
            print("no counters:", e)


if __name__ == "__main__":
    main()
