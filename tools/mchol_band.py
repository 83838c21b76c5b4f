"""Per-task stamps of the one-launch Cholesky, one emulator, by ROW BAND: for the rows 2j+4, 2j+5 (the band that becomes the chain of
block column j+1 and the diagonal block j+2) every task T(r, c) with its stamps -- drawn, last operand wait ends, GEMM done, tile written,
pack seen, done -- so that one can see what the chain tasks of a late block column were waiting for.
usage: mchol_band.py trace.bin [slot=0] [first band=6] [last band=10]     (trace.bin: one launch, as tools/jobs/r3_tr.sh cuts it)"""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64)
nb, ntasks, NP, g = [int(x) for x in raw[:4].astype(np.int64)]
trw = g // 1000000 if g >= 1000000 else 8
tr = raw[4:4 + nb * ntasks * trw].reshape(nb, ntasks, trw).astype(np.int64)
slot = int(sys.argv[2]) if len(sys.argv) > 2 else 0
j0, j1 = (int(sys.argv[3]) if len(sys.argv) > 3 else 6), (int(sys.argv[4]) if len(sys.argv) > 4 else 10)
t0 = int(tr[:, :, 0][tr[:, :, 0] > 0].min())
us = lambda x: (int(x) - t0) / 100.0 if x else -1.0
w = tr[0, :, 7] & 0xffffffff
typ, col, row = (w >> 30) & 3, (w >> 15) & 0x3fff, w & 0x7fff
pos = {(int(typ[p]), int(col[p]), int(row[p]) if typ[p] else 0): p for p in range(ntasks)}
K = NP // 128
for j in range(j0, j1 + 1):
    print("band %d (rows %d, %d):" % (j, 2 * j + 4, 2 * j + 5))
    for c in range(max(0, j - 3), min(K, j + 2)):
        for r in (2 * j + 4, 2 * j + 5):
            p = pos.get((2, c, r))
            if p is None: continue
            T = tr[slot, p]
            role = {j - 1: "below pair", j: "pair", j + 1: "chain"}.get(c, "bulk")
            print("   T(%2d,%2d) %-10s queue pos %4d  drawn %7.1f  operands %7.1f  gemm %7.1f  written %7.1f  pack seen %7.1f  done %7.1f   waited %5.1f" % (
                r, c, role, p, us(T[0]), us(T[2]), us(T[3]), us(T[4]), us(T[8]), us(T[5]), T[9] / 100.0))
    for c in (j + 1, j + 2):
        p = pos.get((0, c, 0))
        if p is not None:
            D = tr[slot, p]
            print("   D(%2d)               queue pos %4d  drawn %7.1f  inputs   %7.1f  done %7.1f" % (c, p, us(D[0]), us(D[2]), us(D[5])))
