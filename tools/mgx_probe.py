"""ctypes binding of tools/probe/libmgx_probe.so: the clock / memory / instruction-fetch probes.

Measurement aid (bench.py's ``gpu_state``, tools/gpu_state.py, tools/box_class.py); not part of the product and
not behind include/mgx.h.  Built in place on first use (tools/probe/build.py)."""

import ctypes
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def library():
    global _lib
    if _lib is None:
        spec = importlib.util.spec_from_file_location("mgx_probe_build", os.path.join(_HERE, "probe", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        try:
            path = mod.build()
        except Exception:                            # noqa: BLE001 -- no hipcc here: use the binary that travelled
            path = mod.OUT
        lib = ctypes.CDLL(path)
        lib.mgx_probe_last_error.restype = ctypes.c_char_p
        lib.mgx_probe_clock.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        lib.mgx_probe_memory.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError("libmgx_probe: " + library().mgx_probe_last_error().decode("utf-8", "replace"))


def clock(device_index, workgroups, iterations):
    """[shader cycles, 100 MHz ticks, shader MHz, kernel ms] of ``workgroups`` x 256 threads of dependent FMAs."""
    out = (ctypes.c_double * 4)()
    _check(library().mgx_probe_clock(int(device_index), int(workgroups), int(iterations), out))
    return list(out)


def memory(device_index):
    """The 14 numbers of mgx_probe_memory (tools/probe/mgx_probe.hip)."""
    out = (ctypes.c_double * 14)()
    _check(library().mgx_probe_memory(int(device_index), out))
    return list(out)
