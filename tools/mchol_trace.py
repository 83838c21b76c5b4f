"""Read the per-task time stamps of the one-launch Cholesky (MOGP_MC_TRACE=<file>, kernels_mchol.hip) and print
  * the dependent chain of one emulator: per block column D(c) -> T(2c+2, c), T(2c+3, c) -> D(c+1),
  * where the workgroups' time went (by task type: waiting, GEMM, write-back, panel solve, diagonal block),
  * the number of busy workgroups over time.
usage: mchol_trace.py file [launch=-1] [slot=0]"""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
launches, off = [], 0
while off < raw.size:
    nb, ntasks, NP, g = [int(x) for x in raw[off:off + 4].astype(np.int64)]
    trw, grid = (g // 1000000, g % 1000000) if g >= 1000000 else (8, g)
    words = nb * ntasks * trw
    launches.append((nb, ntasks, NP, grid, raw[off + 4:off + 4 + words].reshape(nb, ntasks, trw).astype(np.int64)))
    off += 4 + words
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
slot = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nb, ntasks, NP, grid, tr = launches[which]
t0 = int(tr[:, :, 0][tr[:, :, 0] > 0].min())
tend = int(tr[:, :, 5].max())
us = lambda x: (int(x) - t0) / 100.0          # 100 MHz
total_us = (tend - t0) / 100.0
print("# %d launches in file; launch %d: nb=%d NP=%d tasks/emulator=%d grid=%d; kernel %.1f us" % (len(launches), which, nb, NP, ntasks, grid, total_us))
w = tr[0, :, 7] & 0xffffffff
typ, col, row = (w >> 30) & 3, (w >> 15) & 0x3fff, w & 0x7fff
tasks = {}
for p in range(ntasks):
    tasks[(int(typ[p]), int(col[p]), int(row[p]) if typ[p] else 0)] = tr[slot, p]
K = NP // 128
print("col | D: inputs seen -> done (dur) | T0 / T1: pack seen (+lat after D) -> done (dur) | D(c+1) sees the panel (+lat after the later T) | period")
prev = None
for c in range(K):
    D = tasks[(0, c, 0)]
    line = "%3d | D %7.1f ->%7.1f (%5.1f)" % (c, us(D[2]), us(D[5]), us(D[5]) - us(D[2]))
    Ts = [tasks.get((2, c, r)) for r in (2 * c + 2, 2 * c + 3)]
    for i, T in enumerate(Ts):
        if T is not None:
            line += " | T%d %7.1f (+%4.1f) ->%7.1f (%5.1f) [drawn %7.1f ops %7.1f written %7.1f]" % (
                i, us(T[8]), us(T[8]) - us(D[5]), us(T[5]), us(T[5]) - us(T[8]), us(T[0]), us(T[2]) if T[2] else -1, us(T[4]) if T[4] else -1)
    if Ts[0] is not None and (0, c + 1, 0) in tasks:
        tdone = max(us(x[5]) for x in Ts if x is not None)
        line += " | D' +%4.1f" % (us(tasks[(0, c + 1, 0)][2]) - tdone)
    if prev is not None:
        line += " | period %5.1f" % (us(D[2]) - prev)
    prev = us(D[2])
    if K <= 20 or c < 6 or c >= K - 4 or c % 10 == 0:
        print(line)
# where the time went
dur = (tr[:, :, 5] - tr[:, :, 0]) / 100.0
wait = tr[:, :, 9] / 100.0
print("# by task type (all emulators): count, sum of durations, of which waiting; mean duration / wait / last GEMM segment / write-back / solve")
for ty, name in ((0, "D"), (1, "G"), (2, "T"), (3, "TT")):
    m = typ == ty
    if not m.any():
        continue
    d, wv = dur[:, m], wait[:, m]
    x = tr[:, m]
    seg = np.where(x[:, :, 3] > 0, (x[:, :, 3] - x[:, :, 2]) / 100.0, 0.)
    wb = np.where(x[:, :, 4] > 0, (x[:, :, 4] - x[:, :, 3]) / 100.0, 0.)
    sol = np.where(x[:, :, 8] > 0, (x[:, :, 5] - x[:, :, 8]) / 100.0, 0.)
    print("#   %s: %6d tasks, %10.0f us total, %10.0f us waiting (%.0f %%); mean %.1f / %.1f / %.1f / %.1f / %.1f us" % (
        name, d.size, d.sum(), wv.sum(), 100 * wv.sum() / d.sum(), d.mean(), wv.mean(), seg.mean(), wb.mean(), sol.mean()))
    if ty == 2 and x.shape[2] >= 14:
        bm = (x[:, :, 10] > 0) & (x[:, :, 13] > 0)
        if bm.any():
            ph = [((x[:, :, b] - x[:, :, a]) / 100.0)[bm].mean() for a, b in ((8, 10), (10, 11), (11, 12), (12, 13), (13, 5))]
            print("#      bulk solve phases (us): pack seen -> first barrier %.1f | steps 0-3 %.1f | steps 4-7 %.1f | drain %.1f | barrier + publish %.1f   (%d tasks)" % (
                ph[0], ph[1], ph[2], ph[3], ph[4], int(bm.sum())))
    if ty == 2 and x.shape[2] >= 30:
        bm = (x[:, :, 10] > 0) & (x[:, :, 29] > 0)
        if bm.any():
            prev = x[:, :, 10]
            parts = []
            for b in range(8):
                ch = ((x[:, :, 14 + 2 * b] - prev) / 100.0)[bm].mean()
                al = ((x[:, :, 15 + 2 * b] - x[:, :, 14 + 2 * b]) / 100.0)[bm].mean()
                parts.append("%d: %.2f + %.2f" % (b, ch, al))
                prev = x[:, :, 15 + 2 * b]
            print("#      bulk solve per block step (us): wave 0's chain + until all waves are through the step's barrier | " + " | ".join(parts))
    if ty >= 2:
        # GEMM time per 16-deep k-step (all segments: duration - waits - write-back - solve, over 8 c steps), by block column
        cc = col[m]
        gemm = d - wv - wb - sol
        per = [(int(c), float(gemm[:, cc == c].mean() / (8 * c))) for c in sorted(set(cc.tolist())) if c > 0 and (cc == c).any()]
        print("#      GEMM us per k-step by block column: " + " ".join("%d:%.2f" % pc for pc in per))
busy = (dur - wait).sum()
print("# busy (not waiting) workgroup time %.0f us = %.1f of %d workgroups over the kernel's %.1f us" % (busy, busy / total_us, grid, total_us))
# busy workgroups over time (20 bins): a task counts as busy outside its waits -- approximated by its busy fraction
bins = 20
edges = np.linspace(t0, tend, bins + 1)
act = np.zeros(bins)
s_, e_ = tr[:, :, 0].ravel(), tr[:, :, 5].ravel()
frac = np.clip(1 - (wait / np.maximum(dur, 1e-9)).ravel(), 0, 1)
for b in range(bins):
    ov = np.clip(np.minimum(e_, edges[b + 1]) - np.maximum(s_, edges[b]), 0, None)
    act[b] = (ov * frac).sum() / (edges[b + 1] - edges[b])
print("# busy workgroups per 1/20 of the kernel: " + " ".join("%d" % a for a in act))
ids = set(int(v) & 0xffff for v in tr[:, :, 6].ravel())
print("# distinct (XCC_ID, HW_ID[15:8]) values seen: %d" % len(ids))
