"""What clock and power does the GPU hold under the fp64 matrix-core kernels?  Samples the amdgpu hwmon files (freq1_input = shader clock,
power1_average / power1_input) -- and, when they are missing, `rocm-smi --showclocks --showpower` -- every 50 ms in a thread while a workload runs
for SECS seconds: the predictive-variance GEMM (64 x 10^4 points, n = 2000), the one-launch Cholesky (64 x n = 2000), the covariance build, idle.
usage: python tools/clock_watch.py [SECS=4]"""
import glob
import os
import re
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M                        # noqa: E402
from mogp_emulator_amd.Priors import GPPriors        # noqa: E402
from bench import synth                              # noqa: E402

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0


def hwmon_files():
    """hwmon directory of the GPU this process computes on (a box shows every GPU of its host in sysfs): matched by PCI address"""
    import torch
    pr = torch.cuda.get_device_properties(0)
    want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1), getattr(pr, "pci_device_id", 0))
    out = {"pci": want}
    for card in glob.glob("/sys/class/drm/card*"):
        try:
            slot = [l.split("=", 1)[1].strip() for l in open(os.path.join(card, "device", "uevent")) if l.startswith("PCI_SLOT_NAME")][0]
        except Exception:
            continue
        if not slot.lower().startswith(want):
            continue
        for h in glob.glob(os.path.join(card, "device", "hwmon", "hwmon*")):
            for key, names in (("sclk_hz", ("freq1_input",)), ("power_uw", ("power1_average", "power1_input")), ("mclk_hz", ("freq2_input",))):
                for nm in names:
                    p = os.path.join(h, nm)
                    if os.path.exists(p) and key not in out:
                        out[key] = p
    return out


FILES = hwmon_files()


def sample():
    s = {}
    for k, p in FILES.items():
        if k == "pci":
            continue
        try:
            s[k] = float(open(p).read().strip())
        except Exception:
            pass
    if "sclk_hz" not in s:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
            if m:
                s["sclk_hz"] = float(m.group(1)) * 1e6
            m = re.search(r"Power \(W\): ([0-9.]+)", o)
            if m:
                s["power_uw"] = float(m.group(1)) * 1e6
        except Exception:
            pass
    return s


def watch(name, fn):
    stop, rows = threading.Event(), []

    def poll():
        while not stop.is_set():
            rows.append(sample())
            time.sleep(0.05)

    th = threading.Thread(target=poll)
    fn()                                  # warm
    th.start()
    t0, it = time.perf_counter(), 0
    while time.perf_counter() - t0 < SECS:
        fn()
        it += 1
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    rows = rows[len(rows) // 4:]          # the settled part
    line = "%-34s %7.3f ms per call" % (name, dt / max(it, 1) * 1e3)
    for key, unit, sc in (("sclk_hz", "MHz", 1e-6), ("power_uw", "W", 1e-6), ("mclk_hz", "MHz (memory)", 1e-6)):
        v = [r[key] * sc for r in rows if key in r]
        if v:
            line += "   %s mean %.0f min %.0f max %.0f" % (unit, np.mean(v), np.min(v), np.max(v))
    print(line, flush=True)


print("hwmon files:", FILES or "none (rocm-smi)")
for nm in ("power1_cap", "power1_cap_default", "power1_cap_max"):
    if "power_uw" in FILES:
        pth = os.path.join(os.path.dirname(FILES["power_uw"]), nm)
        if os.path.exists(pth):
            try:
                print("%s = %.0f W" % (nm, float(open(pth).read().strip()) * 1e-6))
            except Exception:
                pass
B, n, d, m = 64, 2000, 10, 10000
X, T, Xs = synth(2, n, d, B, m)
gp = M.MultiOutputGP_GPU(X, T, nugget=1e-6, priors=GPPriors(n_corr=d, nugget_type="fixed"))
theta = np.tile(np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.]), (B, 1))
gp.fit(theta)
mo = gp._mogp_gpu
import torch                                         # noqa: E402
dXs = torch.as_tensor(Xs, device="cuda")
watch("idle (sleep)", lambda: time.sleep(0.2))
watch("predict (variance GEMM 92 %)", lambda: (gp.predict(Xs, unc=True, deriv=False), None)[1])
watch("fit (Cholesky 86 %)", lambda: mo.eval(theta, grad=False))
watch("fit + gradient (L^-1, K^-1 GEMMs)", lambda: mo.eval(theta, grad=True))
watch("idle (sleep)", lambda: time.sleep(0.2))
