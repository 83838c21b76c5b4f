import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
print(cols)
rows = con.execute("select start, end, name, grid_x, workgroup_x from kernels order by start").fetchall() if 'grid_x' in cols else None
if rows is None:
    rows = con.execute("select start, end, name, grid_size_x, workgroup_size_x from kernels order by start").fetchall() if 'grid_size_x' in cols else con.execute("select start, end, name, 0, 0 from kernels order by start").fetchall()
# take last fit: find last cov_build
idx = [i for i, r in enumerate(rows) if 'cov_build' in r[2]]
seg = rows[idx[-2]:idx[-1]]
ups = [r for r in seg if 'update_kernel<2' in r[2]]
print(len(ups))
NP = 2048
tot = 0
# sequence per step o=128..1920: pair(o) then U3(o) ; first step o=0 has only U3
k = 0
import itertools
o = 0
out = []
i = 0
# first launch is U3 for o=0
for r in ups:
    dur = (r[1] - r[0]) / 1e3
    wgs = r[3] // max(r[4], 1) if r[4] else 0
    out.append((dur, wgs))
# reconstruct: launches in order: U3(0), pair(128), U3(128), pair(256), ...
seq = [('U3', 0)]
for oo in range(128, 2048, 128):
    seq.append(('pair', oo)); seq.append(('U3', oo))
for (kind, oo), (dur, wgs) in zip(seq, out):
    if kind == 'pair':
        nt = (NP - oo) // 64; tiles = 64 * (2 * nt - 1); K = oo
    else:
        nt = (NP - oo - 64) // 64; tiles = 64 * nt; K = 64
    fl = tiles * 64 * 64 * K * 2
    print("%5s o=%4d tiles %5d (%.2f rounds of 1024) K %4d  %7.1f us  %5.1f TF" % (kind, oo, tiles, tiles / 1024, K, dur, fl / dur / 1e6 if dur else 0))
