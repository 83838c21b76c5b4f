"""GPU busy time (union of kernel intervals) vs wall time from a rocprofv3 --kernel-trace database."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select start, end, name from kernels order by start").fetchall()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[int(len(rows) * skip / 100):]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in rows)
print("wall %.2f ms, busy (union) %.2f ms (%.1f %%), sum of kernel times %.2f ms, kernels %d" % ((t1 - t0) / 1e6, busy / 1e6, 100. * busy / (t1 - t0), tot / 1e6, len(rows)))
