"""Timeline of one evaluation from a rocprofv3 trace: every kernel and memory copy of the LAST evaluation with its start (relative, us), duration and
the gap to the previous command.   usage: python tools/timeline.py <dir with *_kernel_trace.csv [and *_memory_copy_trace.csv]> [first kernel name part]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "cov_build"
rows = []
for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Name", ""))))
rows.sort()
starts = [i for i, r in enumerate(rows) if first in r[2]]
if len(starts) < 3:
    print("no evaluations found", len(rows)); sys.exit(1)
a, b = starts[-2], starts[-1]
# commands of the last-but-one evaluation: from shortly before its first kernel to the next evaluation's first kernel
lo = a
while lo > 0 and rows[a][0] - rows[lo - 1][1] < 60000:      # commands within 60 us in front of the K build belong to this evaluation
    lo -= 1
t0, prev_end = rows[lo][0], None
for s, e, name in rows[lo:b]:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.1f us  +%7.1f dur  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, name))
    prev_end = e
print("evaluation period (first kernel to first kernel): %.1f us" % ((rows[b][0] - rows[a][0]) / 1e3))
