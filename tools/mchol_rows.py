"""Round 6: per-task stamps of the LAST block columns of one emulator (MOGP_MC_TRACE file): for every T task of the columns c >= C0 the times it
was drawn, got its last operand, ended its GEMM, saw the pack, finished -- to see where the chain of the tail waits.
usage: mchol_rows.py file [launch=-2] [slot=0] [C0=9]"""
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
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
slot = int(sys.argv[3]) if len(sys.argv) > 3 else 0
C0 = int(sys.argv[4]) if len(sys.argv) > 4 else 9
nb, ntasks, NP, grid, tr = launches[which]
t0 = int(tr[:, :, 0][tr[:, :, 0] > 0].min())
us = lambda x: (int(x) - t0) / 100.0 if x else -1.
w = tr[0, :, 7] & 0xffffffff
typ, col, row = (w >> 30) & 3, (w >> 15) & 0x3fff, w & 0x7fff
pos = (tr[slot, :, 7] >> 32)
print("# launch %d slot %d: nb=%d NP=%d grid=%d" % (which, slot, nb, NP, grid))
print("type col row | ticket | drawn  last-wait-begin  operands  gemm-end  written  pack-seen  done | waits(us) | hw")
order = np.argsort(tr[slot, :, 0])
for p in order:
    if col[p] < C0:
        continue
    x = tr[slot, p]
    print("%s %3d %3d | %5d | %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f | %6.1f | %x" % (
        "DGT"[typ[p]], col[p], row[p] if typ[p] else 0, pos[p], us(x[0]), us(x[1]), us(x[2]), us(x[3]), us(x[4]), us(x[8]), us(x[5]), x[9] / 100.0, x[6] & 0xffff))
