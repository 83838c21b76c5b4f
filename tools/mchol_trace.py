"""Read the per-task time stamps of the one-launch Cholesky (MOGP_MC_TRACE=<file>, kernels_mchol.hip) and print the dependent
chain of one emulator: per block column D(c) -> T(2c+2, c), T(2c+3, c) -> G(2c+2, c+1), G(2c+3, c+1) -> D(c+1), with the time
each task spent working / waiting and the hand-off latencies between them.  usage: mchol_trace.py file [launch=-1] [slot=0]"""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
launches, off = [], 0
while off < raw.size:
    nb, ntasks, NP, grid = [int(x) for x in raw[off:off + 4].astype(np.int64)]
    words = nb * ntasks * 8
    launches.append((nb, ntasks, NP, grid, raw[off + 4:off + 4 + words].reshape(nb, ntasks, 8)))
    off += 4 + words
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
slot = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nb, ntasks, NP, grid, tr = launches[which]
t0 = int(tr[:, :, 0][tr[:, :, 0] > 0].min())
us = lambda x: (int(x) - t0) / 100.0          # 100 MHz
print("# %d launches in file; launch %d: nb=%d NP=%d tasks/emulator=%d grid=%d; total %.1f us" % (
    len(launches), which, nb, NP, ntasks, grid, (int(tr[:, :, 5].max()) - t0) / 100.0))
tasks = {}
for p in range(ntasks):
    w = int(tr[slot, p, 7]) & 0xffffffff
    typ, c, r = (w >> 30) & 3, (w >> 15) & 0x7fff, w & 0x7fff
    tasks[(typ, c, r if typ else 0)] = tr[slot, p]
K = NP // 128
print("col |  D: ready  start->done (dur) | T0: pack seen (+lat) done (dur) | T1 done | G0: op seen (+lat) done (dur) | G1 done | D(c+1) sees tiles (+lat) | column period")
prev = None
for c in range(K):
    D = tasks[(0, c, 0)]
    line = "%3d | D wait %7.1f..%7.1f run ->%7.1f (%5.1f)" % (c, us(D[1]), us(D[2]), us(D[5]), us(D[5]) - us(D[2]))
    for r in (2 * c + 2, 2 * c + 3):
        T = tasks.get((2, c, r))
        if T is not None:
            line += " | T%d pack@%7.1f (+%4.1f) ->%7.1f (%5.1f)" % (r - 2 * c - 2, us(T[4]), us(T[4]) - us(D[5]), us(T[5]), us(T[5]) - us(T[4]))
    Ts = [tasks.get((2, c, r)) for r in (2 * c + 2, 2 * c + 3)]
    if Ts[0] is not None:
        tdone = max(us(x[5]) for x in Ts if x is not None)
        Dn = tasks.get((0, c + 1, 0))
        if Dn is not None:
            line += " | D' sees panel +%4.1f" % (us(Dn[2]) - tdone)
            G = [tasks.get((1, c + 1, r)) for r in (2 * c + 2, 2 * c + 3)]
            if G[0] is not None:
                line += " (G done %7.1f, %7.1f)" % (us(G[0][5]), us(G[1][5]))
    if prev is not None:
        line += " | period %5.1f" % (us(D[2]) - prev)
    prev = us(D[2])
    print(line)
ids = set()
cnt = {}
for z in range(nb):
    for p in range(ntasks):
        ids.add(int(tr[z, p, 6]) & 0xffff)
print("# distinct (XCC_ID, HW_ID[15:8]) values seen: %d" % len(ids))
# how busy were the workers: sum of task durations minus waits
busy = 0.0
for z in range(nb):
    for p in range(ntasks):
        x = tr[z, p]
        w = (int(x[2]) - int(x[1]) if x[2] and x[1] else 0) + (int(x[4]) - int(x[3]) if x[4] and x[3] else 0)
        busy += (int(x[5]) - int(x[0]) - w) / 100.0
print("# sum over tasks of (duration - last waits): %.1f us = %.1f workgroup-equivalents over the launch" % (busy, busy / ((int(tr[:, :, 5].max()) - t0) / 100.0)))
