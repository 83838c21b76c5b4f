"""
Counterpart of mogp_emulator/LibGPGPU.py:1-14: import shim around the native module.
Here the native module is ``mogp_emulator_amd.libgpgpu`` (ctypes over libmogp_hip.so).
"""
HAVE_LIBGPGPU = False
_IMPORT_ERROR = None
try:
    from .libgpgpu import *          # noqa: F401,F403
    from .libgpgpu import have_compatible_device
    HAVE_LIBGPGPU = True
except (OSError, AttributeError, ImportError) as exc:    # library not built / symbol missing
    _IMPORT_ERROR = exc


def gpu_usable():
    """True when the library loaded AND a gfx950 device is visible (LibGPGPU.py:13-14)."""
    return HAVE_LIBGPGPU and bool(have_compatible_device())
