"""File the outputs of tools/jobs/r5_final.sh (gpurun_out/final_<commit>/) under profiles/<round>_<commit>_* (round: env ROUND, default r05) with explanatory
headers and derive profiles/<round>_traffic.json (what bench.py reports as roofline.traffic).  usage: collect_profiles.py <commit>
REFUSES when the library that ran the job was not built from <commit> (mogp_emulator_amd/libmogp_hip.build, written by the Makefile), or when HEAD differs
from <commit> in anything the job executed (library, package, bench.py, tools)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
RND = os.environ.get("ROUND", "r05")
src = os.path.join(ROOT, "gpurun_out", "final_" + tag)
dst = os.path.join(ROOT, "profiles")
head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=7", "HEAD"], capture_output=True, text=True).stdout.strip()
built = open(os.path.join(src, "build_commit.txt")).read().strip()
# HEAD may have moved on from the commit the job ran on -- but only by commits that leave everything the job executed untouched
# (library sources, headers, the Python package, bench.py, the tools the job calls): documentation, tests, profiles
EXECUTED = ["mogp_emulator_amd", "include", "bench.py", "__graft_entry__.py", "oracle", "tools"]
same_code = tag == built and subprocess.run(["git", "-C", ROOT, "diff", "--quiet", built, "HEAD", "--"] + EXECUTED).returncode == 0
if not (tag == head == built) and not same_code and "--force" not in sys.argv:
    sys.exit("refusing: argument %s, HEAD %s, library build stamp %s must agree, or HEAD must differ from the build commit by documentation / "
             "tests / profiles only (commit, rebuild, run the job, then collect)" % (tag, head, built))


def rd(name):
    with open(os.path.join(src, name)) as fh:
        return fh.read()


def wr(name, text):
    with open(os.path.join(dst, "%s_%s_%s" % (RND, tag, name)), "w") as fh:
        fh.write(text)


def fetch_row(txt, kernel):
    row = [l for l in txt.splitlines() if kernel in l][0].split()
    return float(row[-4]), float(row[-3]), int(row[-1])          # FETCH_SIZE KB, WRITE_SIZE KB, calls


wr("gpu_tests.txt", "# python -m pytest tests -m gpu -x -q --durations=10 at commit %s (tail), then smoke()\n" % tag + rd("gpu_tests.txt"))
wr("bench.json", [l for l in rd("bench.json").strip().splitlines() if l.startswith("{")][-1] + "\n")
wr("kernel_stats.txt", rd("kernel_stats.txt"))
for c in ("C2", "S8", "C4", "C5"):
    wr(c.lower() + "_kernel_stats.txt", rd(c + "_kernel_stats.txt"))
for t in ("1_2000_10", "8_2000_10", "64_2000_10", "1_16000_8"):
    wr("mchol_trace_%s.txt" % t, "# MOGP_MC_TRACE=... CONFIGS=%s python tools/mchol_check.py; python tools/mchol_trace.py (per-task time stamps of the one-launch Cholesky)\n" % t.replace("_", ":")
       + rd("mchol_trace_%s.txt" % t))
wr("mchol_notraffic.txt", "# the one-launch Cholesky at 64 x n=2000 with its GEMM tasks re-reading their first 64 operand columns from the caches (MOGP_MC_NOTRAFFIC=1: same\n"
   "# instruction stream, no fabric traffic, garbage results) against the real thing: python tools/mchol_time.py\n" + rd("mchol_notraffic.txt"))
# FETCH / WRITE
per_launch = None
for suffix, what in (("", "64 x n=2000 x d=10, m=10000"), ("_S8", "the 8-emulator shard: 8 x n=2000, m=10000"), ("_C2", "C2: one n=2000 emulator, m=10000")):
    txt = rd("pmc_fetch_write_kb%s.txt" % suffix)
    lines = []
    for kern in ("predict_var_w_kernel", "mchol_kernel", "kinv_kernel", "cov_build_kernel"):
        try:
            f, w, c = fetch_row(txt, kern)
        except (IndexError, ValueError):
            continue
        gb = (2 * f + w) * 1024 / c / 1e9
        lines.append("#   %-22s (2 x %.4g + %.4g) KB x 1024 / %d launches = %.2f GB per launch\n" % (kern, f, w, c, gb))
        if kern == "predict_var_w_kernel" and suffix == "":
            per_launch = gb * 1e9
    hdr = ("# commit %s: L2-miss traffic per kernel, two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB) over tools/pmc_step.py (%s;\n"
           "#   two fit+gradient evaluations and two predictions; bash tools/pmc_fetch.sh).  gfx950: FETCH_SIZE reports half of the bytes of wide coalesced\n"
           "#   reads (MI355X_MICROARCH.md, HBM section) -> HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE:\n" % (tag, what)) + "".join(lines)
    wr("pmc_fetch_write_kb%s.txt" % suffix, hdr + txt)
with open(os.path.join(dst, RND + "_traffic.json"), "w") as fh:
    json.dump({"predict_var": {"traffic_bytes_per_launch": per_launch,
                               "source": "profiles/%s_%s_pmc_fetch_write_kb.txt: (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 / launches, separate --pmc passes "
                                         "(bash tools/pmc_fetch.sh), gfx950 FETCH_SIZE x2 correction" % (RND, tag),
                               "config": "64 outputs n=2000 d=10 m=10000, one launch of 10112 padded points per predict (MOGP_KS_BUDGET_GB=12)"}}, fh)
# SQ
for B in (64, 8, 1):
    txt = rd("pmc_sq_B%d.txt" % B)
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
    hdr = ("# commit %s, %d x n=2000: rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY\n"
           "#   SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -- python tools/pmc_step.py (PMC_M=10000; its own run, no trace domains; sums over all dispatches of a kernel)\n"
           "# Normalisation: MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 1024 SIMDs):\n" % (tag, B))
    for k, v in sorted(busy.items(), key=lambda kv: -kv[1]):
        if v > 0:
            hdr += "#   %-48s %.3f\n" % (k[:48], v)
    wr("pmc_sq_B%d.txt" % B, hdr + txt)
wr("pmc_tcc.txt", "# commit %s, 64 x n=2000: rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -- python tools/pmc_step.py (own pass): L2 hit rate = HIT / (HIT + MISS)\n" % tag + rd("pmc_tcc.txt"))
if os.path.exists(os.path.join(src, "pmc_sq_valu_B64.txt")):
    wr("pmc_sq_valu_B64.txt", "# commit %s, 64 x n=2000 x d=10, m=10000: rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE\n"
       "#   -- python tools/pmc_step.py (own pass): vector-ALU instructions per kernel (sums over two fit+gradient evaluations and two predictions)\n" % tag + rd("pmc_sq_valu_B64.txt"))
if os.path.exists(os.path.join(src, "gemm_loop_probe.txt")):
    wr("gemm_loop_probe.txt", "# commit %s: tools/gemm_loop_probe.bin 64 40 (the GEMM main loops of the Cholesky tasks alone)\n" % tag + rd("gemm_loop_probe.txt"))
print("profiles/%s_%s_* written; predict_var traffic %.1f GB per launch" % (RND, tag, per_launch / 1e9))
