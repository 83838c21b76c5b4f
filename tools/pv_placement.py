"""Round 6: the predictive variance under different PLACEMENTS of its tiles -- super-tile = 2^lgc column tiles x 64 / 2^lgc row-tile pairs
dealt to consecutive workgroups of one XCD (MOGP_PV_LGC, read once per process: run this script once per value), both kernels (MOGP_PV_Q) -- with the
kernel's HIP-event time, shader clock and board power (hwmon of the GPU the process runs on, sampled every 50 ms) side by side.  Traffic: tools/pmc_fetch.sh
with the same environment.  usage: MOGP_PV_LGC=<k> [MOGP_PV_Q=1] python tools/pv_placement.py [SECS=3]"""
import ctypes, glob, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogp_emulator_amd as M
from mogp_emulator_amd import _capi
from mogp_emulator_amd.Priors import GPPriors
from bench import synth
import torch
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
lib = _capi.load()


def hwmon():
    pr = torch.cuda.get_device_properties(0)
    want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1), getattr(pr, "pci_device_id", 0))
    out = {}
    for card in glob.glob("/sys/class/drm/card*"):
        try:
            slot = [l.split("=", 1)[1].strip() for l in open(os.path.join(card, "device", "uevent")) if l.startswith("PCI_SLOT_NAME")][0]
        except Exception:
            continue
        if not slot.lower().startswith(want):
            continue
        for h in glob.glob(os.path.join(card, "device", "hwmon", "hwmon*")):
            for key, names in (("MHz", ("freq1_input",)), ("W", ("power1_average", "power1_input"))):
                for nm in names:
                    p = os.path.join(h, nm)
                    if os.path.exists(p) and key not in out:
                        out[key] = p
    return out


FILES = hwmon()
B, n, d, m = 64, 2000, 10, 10000
X, T, Xs = synth(2, n, d, B, m)
gp = M.MultiOutputGP_GPU(X, T, nugget=1e-6, priors=GPPriors(n_corr=d, nugget_type="fixed"))
theta = np.tile(np.array([-2. * np.log(0.3 * np.sqrt(d))] * d + [0.]), (B, 1))
mo = gp._mogp_gpu
mo.eval(theta, grad=True)
d_Xs = torch.from_numpy(Xs).cuda(); d_mean = torch.empty((B, m), dtype=torch.float64, device="cuda"); d_var = torch.empty_like(d_mean)
call = lambda: mo.predict_variance_batch_dev(d_Xs.data_ptr(), m, d_mean.data_ptr(), d_var.data_ptr())
call(); call()
stop, rows = threading.Event(), []


def poll():
    while not stop.is_set():
        s = {}
        for k, p in FILES.items():
            try:
                s[k] = float(open(p).read().strip()) * 1e-6
            except Exception:
                pass
        rows.append(s)
        time.sleep(0.05)


lib.mogp_profile_reset(); lib.mogp_profile_enable(1)
th = threading.Thread(target=poll); th.start()
t0, it = time.perf_counter(), 0
while time.perf_counter() - t0 < SECS:
    call(); it += 1
stop.set(); th.join()
lib.mogp_profile_enable(0)
ms, cnt, fl, by = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double()
lib.mogp_profile_get(b"predict_var", ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl), ctypes.byref(by))
rows = rows[len(rows) // 4:]
line = "MOGP_PV_LGC=%s MOGP_PV_Q=%s: predict_var %.3f ms = %.2f TFLOP/s (%d launches)" % (
    os.environ.get("MOGP_PV_LGC", "default(3)"), os.environ.get("MOGP_PV_Q", "default"), ms.value / max(cnt.value, 1), fl.value / max(ms.value, 1e-9) * 1e-9, cnt.value)
for k in ("MHz", "W"):
    v = [r[k] for r in rows if k in r]
    if v:
        line += "   %s mean %.0f min %.0f max %.0f" % (k, np.mean(v), np.min(v), np.max(v))
print(line + "   checksum %.12g" % float(d_var.sum().item()), flush=True)
