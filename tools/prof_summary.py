"""Summarise a rocprofv3 --kernel-trace results.db into a per-kernel table (name, calls, total/avg/min/max)."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                   "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("# %s\n# total kernel time %.3f ms" % (" ".join(sys.argv[2:]), tot))
print("%-100s %7s %11s %11s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
for r in rows:
    print("%-100s %7d %11.3f %11.1f %10.1f %10.1f %5.1f%%" % (r[0][:100], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
