"""Per-kernel PMC summary from rocprofv3 --pmc result dbs (sum over dispatches, and per-dispatch average)."""
import sqlite3, sys, collections
def load(f):
    con = sqlite3.connect(f)
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for k, c, v in con.execute("select kernel_name, counter_name, value from counters_collection"):
        k = k.split("(")[0].replace("void ", "")
        out[k][c] += v; cnt[k][c] += 1
    return out, cnt
allc = collections.defaultdict(dict); calls = {}
for f in sys.argv[1:]:
    o, c = load(f)
    for k in o:
        for cn in o[k]:
            allc[k][cn] = o[k][cn]; calls[(k, cn)] = c[k][cn]
names = sorted({cn for k in allc for cn in allc[k]})
print("kernel".ljust(48), " ".join(n[:18].rjust(18) for n in names))
for k in sorted(allc, key=lambda k: -allc[k].get("SQ_BUSY_CYCLES", allc[k].get("FETCH_SIZE", 0))):
    print(k[:48].ljust(48), " ".join(("%.4g" % allc[k].get(n, float('nan'))).rjust(18) for n in names), " calls", calls.get((k, names[0]), 0))
