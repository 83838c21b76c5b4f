"""Copy the outputs of tools/jobs/r3_final.sh (gpurun_out/final_<tag>/) into profiles/r03_<tag>_* with explanatory headers and
derive profiles/r03_traffic.json (what bench.py reports as roofline.traffic).  usage: collect_profiles.py <tag>"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "final_" + tag)
dst = os.path.join(ROOT, "profiles")


def rd(name):
    with open(os.path.join(src, name)) as fh:
        return fh.read()


def wr(name, text):
    with open(os.path.join(dst, "r03_%s_%s" % (tag, name)), "w") as fh:
        fh.write(text)


wr("gpu_tests.txt", "# python -m pytest tests -m gpu -x -q --durations=10 at commit %s (tail)\n" % tag + rd("gpu_tests.txt"))
line = [l for l in rd("bench.json").strip().splitlines() if l.startswith("{")][-1]
wr("bench.json", line + "\n")
wr("kernel_stats.txt", rd("kernel_stats.txt"))
for c in ("C4", "C5"):
    wr(c.lower() + "_kernel_stats.txt", rd(c + "_kernel_stats.txt"))
for t in ("8_2000_10", "64_2000_10", "1_16000_8"):
    wr("mchol_trace_%s.txt" % t, "# MOGP_MC_TRACE=... CONFIGS=%s python tools/mchol_check.py; python tools/mchol_trace.py (per-task time stamps of the one-launch Cholesky)\n" % t.replace("_", ":")
       + rd("mchol_trace_%s.txt" % t))
# FETCH / WRITE
txt = rd("pmc_fetch_write_kb.txt")
row = [l for l in txt.splitlines() if "predict_var_w_kernel" in l][0].split()
fetch_kb, write_kb, calls = float(row[-4]), float(row[-3]), int(row[-1])
per_launch = (2 * fetch_kb + write_kb) * 1024 / calls
hdr = ("# commit %s: L2-miss traffic per kernel, two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB) over tools/pmc_step.py\n"
       "#   (bash tools/pmc_fetch.sh; 64 x n=2000 x d=10, m=10000, two fit+gradient evaluations and two predictions, K* chunk 12 GB = one\n"
       "#   predictive-variance launch of 10112 padded points per prediction).  gfx950: FETCH_SIZE reports half of the bytes of wide coalesced\n"
       "#   reads (MI355X_MICROARCH.md, HBM section) -> HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE.\n"
       "#   predict_var_w_kernel<2,4,true>: (2 x %.4g + %.4g) KB x 1024 / %d launches = %.1f GB per launch\n"
       "#   mchol_kernel (the whole Cholesky of the batch in one launch): see the row below; algorithmic 4 n^2 B per emulator = 1.07 GB.\n" % (
           tag, fetch_kb, write_kb, calls, per_launch / 1e9))
wr("pmc_fetch_write_kb.txt", hdr + txt)
with open(os.path.join(dst, "r03_traffic.json"), "w") as fh:
    json.dump({"predict_var": {"traffic_bytes_per_launch": per_launch,
                               "source": "profiles/r03_%s_pmc_fetch_write_kb.txt: (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 / launches, separate --pmc passes "
                                         "(bash tools/pmc_fetch.sh), gfx950 FETCH_SIZE x2 correction; WRITE_SIZE of this kernel is scratch traffic of the phase "
                                         "changes (84 B per thread)" % tag,
                               "config": "64 outputs n=2000 d=10 m=10000, one launch of 10112 padded points per predict (MOGP_KS_BUDGET_GB=12)"}}, fh)
if os.path.exists(os.path.join(src, "pmc_fetch_write_kb_pv_sync.txt")):
    t2 = rd("pmc_fetch_write_kb_pv_sync.txt")
    r2 = [l for l in t2.splitlines() if "predict_var_w_kernel" in l][0].split()
    f2, w2, c2 = float(r2[-4]), float(r2[-3]), int(r2[-1])
    wr("pmc_fetch_write_kb_pv_sync.txt",
       "# commit %s: the same two passes with MOGP_PV_SYNC=1000 (predictive variance as persistent workgroups in soft lock-step, short pass\n"
       "#   downward): (2 x %.4g + %.4g) KB x 1024 / %d launches = %.1f GB per launch (default above: %.1f GB).  Time of the predict phase,\n"
       "#   default / lock-step alternating in one job (tools/ab.py):\n%s" % (tag, f2, w2, c2, (2 * f2 + w2) * 1024 / c2 / 1e9, per_launch / 1e9,
                                                                    "".join("#   " + l + "\n" for l in rd("pv_sync_ab.txt").strip().splitlines())) + t2)
# SQ
txt = rd("pmc_sq.txt")
busy = {}
for l in txt.splitlines()[1:]:
    f = l.split()
    if len(f) > 9 and f[0].startswith("mogp::"):
        name = " ".join(f[:-10])
        try:
            vals = [float(x) for x in f[-10:-2]]
        except ValueError:
            continue
        if vals[0] > 0:
            busy[name] = vals[3] / (vals[0] / 8 * 1024)
hdr = ("# commit %s: rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY\n"
       "#   SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -- python tools/pmc_step.py (PMC_M=10000; its own run, no trace domains; sums over all dispatches of a kernel)\n"
       "# Normalisation: MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 1024 SIMDs):\n" % tag)
for k, v in sorted(busy.items(), key=lambda kv: -kv[1]):
    if v > 0:
        hdr += "#   %-48s %.3f\n" % (k[:48], v)
wr("pmc_sq.txt", hdr + txt)
print("profiles/r03_%s_* written; predict_var traffic %.1f GB per launch" % (tag, per_launch / 1e9))
