"""Per-launch duration / rate of the left-looking long-K updates of one fit (single stream), from a rocprofv3 kernel trace."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
rows = con.execute("select start, end, name from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'cov_build' in r[2]]
seg = rows[idx[-2]:idx[-1]]
ups = [r for r in seg if 'update_kernel<2' in r[2]]
NP = 2048
o = 128
tot_t = tot_f = 0
for r in ups:
    dur = (r[1] - r[0]) / 1e3
    nt = (NP - o) // 64; tiles = 64 * (2 * nt - 1); K = o
    fl = tiles * 64 * 64 * K * 2
    print("o=%4d tiles %5d (%.2f rounds of 1024) K %4d  %7.1f us  %5.1f TF" % (o, tiles, tiles / 1024, K, dur, fl / dur / 1e6))
    tot_t += dur; tot_f += fl
    o += 128
print("total %.1f us, %.1f TF" % (tot_t, tot_f / tot_t / 1e6))
others = {}
for r in seg:
    if 'update_kernel<2' in r[2]: continue
    k = r[2].split('(')[0][-40:]
    others[k] = others.get(k, 0) + (r[1] - r[0]) / 1e3
print({k: round(v, 1) for k, v in others.items()})
print("wall of the fit segment %.1f us" % ((seg[-1][1] - seg[0][0]) / 1e3))
