"""Print the kernel timeline of ONE evaluation from a rocprofv3 --kernel-trace results.db: start (us, relative), duration,
stream / queue, grid, kernel name.  Usage: timeline.py results.db [index of the cov_build launch to start from] [count]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select start, end, queue_id, stream_id, grid_x, grid_y, name from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "cov_build" in r[6]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 200
i0 = starts[k]
i1 = starts[k + 1] if k + 1 < len(starts) else len(rows)
t0 = rows[i0][0]
busy_end = t0
for r in rows[i0:min(i1, i0 + cnt)]:
    gap = (r[0] - busy_end) / 1e3
    busy_end = max(busy_end, r[1])
    print("%9.1f us  +%7.1f us  gap %6.1f  q%-3s s%-3s grid %6d x %-4d %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, gap, r[2], r[3], r[4], r[5],
                                                                      r[6].replace("mogp::", "").replace("void ", "")[:60]))
print("evaluation wall (first start to last end): %.1f us" % ((max(r[1] for r in rows[i0:i1]) - t0) / 1e3))
