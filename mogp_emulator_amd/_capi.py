"""
ctypes binding of the C ABI declared in include/mogp_hip.h (libmogp_hip.so).

This is the only place the shared library is touched.  There is NO CPU fallback:
if the library cannot be loaded, ``load()`` raises, and every higher layer
(``libgpgpu``) fails loudly instead of silently computing somewhere else.
"""
import ctypes
import os
import struct
from ctypes import POINTER, c_char_p, c_double, c_int, c_longlong, c_uint, c_ulonglong, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MOGP_LIB_PATH: another build of the same library (A/B comparisons of kernel changes, tools/ab.py)
LIB_PATH = os.environ.get("MOGP_LIB_PATH") or os.path.join(_HERE, "libmogp_hip.so")

_lib = None

c_double_p = POINTER(c_double)
c_int_p = POINTER(c_int)


def dptr(a):
    """double* of a C-contiguous float64 ndarray (or NULL for None)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], "need a C-contiguous float64 array"
    return a.ctypes.data_as(c_double_p)


def iptr(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_int_p)


# name -> (restype, argtypes); every symbol of include/mogp_hip.h
SIGNATURES = {
    "mogp_last_error": (c_char_p, []),
    "mogp_have_compatible_device": (c_int, []),
    "mogp_device_count": (c_int, []),
    "mogp_set_device": (c_int, [c_int]),
    "mogp_version": (c_char_p, []),
    "mogp_meanfunc_zero": (c_void_p, []),
    "mogp_meanfunc_fixed": (c_void_p, [c_double]),
    "mogp_meanfunc_const": (c_void_p, []),
    "mogp_meanfunc_poly": (c_void_p, [c_int_p, c_int_p, c_int]),
    "mogp_meanfunc_destroy": (None, [c_void_p]),
    "mogp_meanfunc_n_params": (c_int, [c_void_p]),
    "mogp_meanfunc_mean_f": (c_int, [c_void_p, c_double_p, c_int, c_int, c_double_p, c_int, c_double_p]),
    "mogp_meanfunc_mean_deriv": (c_int, [c_void_p, c_double_p, c_int, c_int, c_double_p, c_int, c_double_p]),
    "mogp_meanfunc_mean_inputderiv": (c_int, [c_void_p, c_double_p, c_int, c_int, c_double_p, c_int, c_double_p]),
    "mogp_densegp_create": (c_void_p, [c_double_p, c_int, c_int, c_double_p, c_uint, c_void_p, c_int, c_int, c_double]),
    "mogp_densegp_create_analytic_mean": (c_void_p, [c_double_p, c_int, c_int, c_double_p, c_uint, c_void_p, c_int, c_int, c_double]),
    "mogp_densegp_set_mean_priors": (c_int, [c_void_p, c_int, c_double_p, c_double_p, c_double_p, c_double]),
    "mogp_densegp_n_beta": (c_int, [c_void_p]),
    "mogp_densegp_get_beta": (c_int, [c_void_p, c_double_p]),
    "mogp_densegp_destroy": (None, [c_void_p]),
    "mogp_densegp_n": (c_int, [c_void_p]),
    "mogp_densegp_D": (c_int, [c_void_p]),
    "mogp_densegp_n_corr": (c_int, [c_void_p]),
    "mogp_densegp_n_params": (c_int, [c_void_p]),
    "mogp_densegp_n_mean": (c_int, [c_void_p]),
    "mogp_densegp_n_data": (c_int, [c_void_p]),
    "mogp_densegp_inputs": (c_int, [c_void_p, c_double_p]),
    "mogp_densegp_targets": (c_int, [c_void_p, c_double_p]),
    "mogp_densegp_theta_fit_status": (c_int, [c_void_p]),
    "mogp_densegp_reset_theta_fit_status": (c_int, [c_void_p]),
    "mogp_densegp_get_theta": (c_int, [c_void_p, c_double_p, c_double_p]),
    "mogp_densegp_create_gppriors": (c_int, [c_void_p, c_int, c_int_p, c_double_p, c_int, c_double_p, c_int, c_double_p]),
    "mogp_densegp_priors_logp": (c_int, [c_void_p, c_double_p, c_int, c_double_p]),
    "mogp_densegp_priors_dlogpdtheta": (c_int, [c_void_p, c_double_p, c_int, c_double_p]),
    "mogp_densegp_priors_sample": (c_int, [c_void_p, c_double_p]),
    "mogp_densegp_fit": (c_int, [c_void_p, c_double_p, c_int]),
    "mogp_densegp_get_logpost": (c_int, [c_void_p, c_double_p, c_int, c_double_p]),
    "mogp_densegp_logpost_deriv": (c_int, [c_void_p, c_double_p, c_int]),
    "mogp_densegp_predict": (c_int, [c_void_p, c_double_p, c_int, c_double_p]),
    "mogp_densegp_predict_variance": (c_int, [c_void_p, c_double_p, c_int, c_double_p, c_double_p]),
    "mogp_densegp_predict_batch": (c_int, [c_void_p, c_double_p, c_int, c_int, c_double_p, c_int]),
    "mogp_densegp_predict_variance_batch": (c_int, [c_void_p, c_double_p, c_int, c_int, c_double_p, c_double_p, c_int]),
    "mogp_densegp_predict_deriv": (c_int, [c_void_p, c_double_p, c_int, c_int, c_double_p, c_int, c_int]),
    "mogp_densegp_predict_full_cov": (c_int, [c_void_p, c_double_p, c_int, c_int, c_double_p, c_double_p]),
    "mogp_densegp_implausibility": (c_int, [c_void_p, c_double_p, c_int, c_int, c_double, c_double, c_double, c_int, c_double_p]),
    "mogp_densegp_loo_variance": (c_int, [c_void_p, c_double_p]),
    "mogp_densegp_get_K": (c_int, [c_void_p, c_double_p]),
    "mogp_densegp_get_invQ": (c_int, [c_void_p, c_double_p]),
    "mogp_densegp_get_invQt": (c_int, [c_void_p, c_double_p]),
    "mogp_densegp_get_cholesky_lower": (c_int, [c_void_p, c_double_p]),
    "mogp_densegp_get_pivot": (c_int, [c_void_p, c_int_p, c_int_p]),
    "mogp_pivot_cholesky": (c_int, [c_double_p, c_int, c_double_p, c_int_p, c_int_p]),
    "mogp_densegp_get_nugget_size": (c_double, [c_void_p]),
    "mogp_densegp_set_nugget_size": (c_int, [c_void_p, c_double]),
    "mogp_densegp_get_nugget_type": (c_int, [c_void_p]),
    "mogp_densegp_set_nugget_type": (c_int, [c_void_p, c_int]),
    "mogp_densegp_get_kernel_type": (c_int, [c_void_p]),
    "mogp_fit_single_GP_MAP": (c_int, [c_void_p, c_int, c_double_p, c_int]),
    "mogp_mogp_create": (c_void_p, [c_double_p, c_int, c_int, c_double_p, c_int, c_uint, c_void_p, c_int, c_int, c_double]),
    "mogp_mogp_create_analytic_mean": (c_void_p, [c_double_p, c_int, c_int, c_double_p, c_int, c_uint, c_void_p, c_int, c_int, c_double]),
    "mogp_mogp_destroy": (None, [c_void_p]),
    "mogp_mogp_n": (c_int, [c_void_p]),
    "mogp_mogp_D": (c_int, [c_void_p]),
    "mogp_mogp_n_emulators": (c_int, [c_void_p]),
    "mogp_mogp_inputs": (c_int, [c_void_p, c_double_p]),
    "mogp_mogp_targets": (c_int, [c_void_p, c_double_p]),
    "mogp_mogp_emulator": (c_void_p, [c_void_p, c_int]),
    "mogp_mogp_get_nugget_type": (c_int, [c_void_p]),
    "mogp_mogp_get_nugget_size": (c_double, [c_void_p]),
    "mogp_mogp_get_fitted_indices": (c_int, [c_void_p, c_int_p]),
    "mogp_mogp_get_unfitted_indices": (c_int, [c_void_p, c_int_p]),
    "mogp_mogp_reset_fit_status": (c_int, [c_void_p]),
    "mogp_mogp_create_priors_for_emulator": (c_int, [c_void_p, c_int, c_int, c_int_p, c_double_p, c_int, c_double_p, c_int, c_double_p]),
    "mogp_mogp_fit": (c_int, [c_void_p, c_double_p, c_int, c_int]),
    "mogp_mogp_fit_emulator": (c_int, [c_void_p, c_int, c_double_p, c_int]),
    "mogp_mogp_eval": (c_int, [c_void_p, c_double_p, c_int, c_int, c_double_p, c_double_p, c_int_p]),
    "mogp_mogp_predict_batch": (c_int, [c_void_p, c_double_p, c_int, c_int, c_double_p]),
    "mogp_mogp_predict_variance_batch": (c_int, [c_void_p, c_double_p, c_int, c_int, c_double_p, c_double_p]),
    "mogp_mogp_predict_deriv": (c_int, [c_void_p, c_double_p, c_int, c_int, c_double_p]),
    "mogp_mogp_implausibility": (c_int, [c_void_p, c_double_p, c_int, c_int, c_double_p, c_double_p, c_double_p, c_int, c_int, c_double_p]),
    "mogp_mogp_predict_full_cov": (c_int, [c_void_p, c_double_p, c_int, c_int, c_double_p, c_double_p]),
    "mogp_mogp_predict_variance_batch_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "mogp_mogp_predict_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mogp_fit_GP_MAP": (c_int, [c_void_p, c_int, c_double_p, c_int]),
    "mogp_set_fit_options": (c_int, [c_int, c_double, c_double, c_ulonglong]),
    "mogp_kernel_eval": (c_int, [c_int, c_int, c_double_p, c_int, c_double_p, c_int, c_int, c_double_p, c_int, c_double_p]),
    "mogp_profile_enable": (c_int, [c_int]),
    "mogp_profile_reset": (c_int, []),
    "mogp_profile_schedule": (c_int, [c_int, c_int]),
    "mogp_profile_get": (c_int, [c_char_p, c_double_p, POINTER(c_longlong), c_double_p, c_double_p]),
    "mogp_profile_counter": (c_int, [c_char_p, POINTER(c_longlong)]),
    "mogp_mchol_task_table": (c_int, [c_int, c_int_p, c_int]),
    "mogp_mchol_task_table_ahead": (c_int, [c_int, c_int_p, c_int]),
    "mogp_dev_malloc": (c_void_p, [c_ulonglong]),
    "mogp_dev_free": (c_int, [c_void_p]),
    "mogp_dev_upload": (c_int, [c_void_p, c_void_p, c_ulonglong]),
    "mogp_dev_download": (c_int, [c_void_p, c_void_p, c_ulonglong]),
    "mogp_dev_synchronize": (c_int, []),
}


def elf_dynamic(path):
    """(DT_SONAME or None, [DT_NEEDED ...]) of a little-endian ELF64 shared object, read with struct -- no external tool.  Only the
    headers, the dynamic section and the strings it points at are read (a HIP runtime is tens of megabytes)."""
    with open(path, "rb") as f:
        def at(off, size):
            f.seek(off)
            return f.read(size)
        ident = at(0, 64)
        if ident[:6] != b"\x7fELF\x02\x01":
            raise OSError("%s is not a little-endian ELF64 file" % path)
        e_phoff, = struct.unpack_from("<Q", ident, 0x20)
        e_phentsize, e_phnum = struct.unpack_from("<HH", ident, 0x36)
        if e_phentsize < 56:
            raise OSError("%s: ELF64 program headers of %d bytes" % (path, e_phentsize))
        ph = at(e_phoff, e_phentsize * e_phnum)
        loads, dyn = [], None
        for k in range(e_phnum):
            p_type, _, p_offset, p_vaddr, _, p_filesz = struct.unpack_from("<IIQQQQ", ph, k * e_phentsize)
            if p_type == 1:
                loads.append((p_vaddr, p_offset, p_filesz))
            elif p_type == 2:
                dyn = (p_offset, p_filesz)
        if dyn is None:
            return None, []
        dsec = at(*dyn)
        entries = [struct.unpack_from("<qQ", dsec, 16 * k) for k in range(len(dsec) // 16)]
        strtab = next((v for t, v in entries if t == 5), None)
        if strtab is None:
            return None, []
        stroff = next((off + strtab - va for va, off, sz in loads if va <= strtab < va + sz), strtab)

        def name(o):
            raw = at(stroff + o, 256)
            return raw[:raw.index(b"\0")].decode()
        soname = next((name(v) for t, v in entries if t == 14), None)
        return soname, [name(v) for t, v in entries if t == 1]


def _bundled_runtime_to_preload(lib_path):
    """Paths of a PyTorch wheel's bundled HSA / HIP runtime to load before `lib_path`, or [] -- only when the bundled
    libamdhip64's DT_SONAME is exactly the libamdhip64.so.N that `lib_path` was linked against (its DT_NEEDED): a wheel built for
    another ROCm major is a different ABI and is left alone (then the system runtime serves this library, as the linker intended)."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return []
    if spec is None or not spec.origin:
        return []
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    hip = os.path.join(libdir, "libamdhip64.so")
    if not os.path.exists(hip):
        return []
    try:
        needed = [n for n in elf_dynamic(lib_path)[1] if n.startswith("libamdhip64.so")]
        soname = elf_dynamic(hip)[0]
    except (OSError, ValueError, IndexError, struct.error):      # a truncated / odd ELF degrades to the system runtime
        return []
    if not needed or soname != needed[0]:
        return []
    hsa = os.path.join(libdir, "libhsa-runtime64.so")
    return ([hsa] if os.path.exists(hsa) else []) + [hip]


def _one_hip_runtime_per_process():
    """A PyTorch-ROCm wheel bundles its own libamdhip64.so.N / libhsa-runtime64.so and loads them by absolute path.  If this
    library has already pulled in the system copies (/opt/rocm), the process ends up with TWO HIP / HSA runtimes, and on some
    hosts the second one finds no device ("No HIP GPUs are available" from torch.cuda after a fit; seen on MI355X boxes in
    round 4).  Loaded the other way round, libmogp_hip.so's NEEDED libamdhip64.so.N binds to the copy that is already there.
    So: when a torch wheel bundles a runtime WITH THE SAME SONAME as the one this library needs (`_bundled_runtime_to_preload`),
    its copy is loaded first -- whichever of the two packages is imported first, there is one runtime (the configuration bench.py
    and the sharded path always ran in).  torch itself is NOT imported.  MOGP_HIP_RUNTIME=system skips this.  Returns the paths
    that were loaded (for the error message of a failing load)."""
    if os.environ.get("MOGP_HIP_RUNTIME", "") == "system":
        return []
    done = []
    for path in _bundled_runtime_to_preload(LIB_PATH):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            done.append(path)
        except OSError:
            break
    return done


def load():
    """Load libmogp_hip.so (once) and attach the prototypes.  Raises OSError /
    AttributeError if the library or one of its symbols is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("libmogp_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "or `make -C mogp_emulator_amd/csrc`" % LIB_PATH)
    preloaded = _one_hip_runtime_per_process()
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        if preloaded:
            raise OSError("%s failed to load after the HIP runtime bundled with PyTorch was loaded first (%s): %s -- "
                          "set MOGP_HIP_RUNTIME=system to bind to the system ROCm runtime instead"
                          % (LIB_PATH, ", ".join(preloaded), e)) from e
        raise
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().mogp_last_error().decode("utf-8", "replace")


def check(status):
    """Non-zero status -> RuntimeError(message), as pybind11 maps std::runtime_error."""
    if status != 0:
        raise RuntimeError(last_error())
