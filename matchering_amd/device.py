"""Device-side plumbing for the Python host: one ``Device`` = one ``mgx_handle``
(one MI355X, one HIP stream).  Buffers are plain HBM allocations addressed by
integer pointers; numpy arrays cross the PCIe boundary only in ``upload`` /
``download``.  No PyTorch: the north star keeps the host a thin ctypes caller.
"""

import ctypes
import threading

import numpy as np

from . import _native
from ._native import MgxReport, check, library


class DeviceBuffer:
    """An HBM allocation owned by a Device (freed on ``release`` or garbage collection)."""

    def __init__(self, device, nbytes):
        self.device = device
        self.nbytes = int(nbytes)
        ptr = ctypes.c_void_p()
        check(library().mgx_malloc(device.handle, self.nbytes, ctypes.byref(ptr)))
        self.ptr = ptr.value

    def release(self):
        if self.ptr and self.device.handle:
            library().mgx_free(self.device.handle, ctypes.c_void_p(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Device:
    """Handle to one GPU.  Raises ``MgxError`` when no GPU is visible: there is no CPU path."""

    def __init__(self, index=0):
        self.index = index
        self.handle = None
        # include/mgx.h: calls on one handle are serialised by the caller.  ctypes drops the GIL while a
        # call blocks, so two threads sharing a Device (the process-wide default one, typically) would
        # otherwise interleave inside the same stream and workspaces.  Every upload -> kernels ->
        # download sequence of this package runs under this lock.
        self.lock = threading.RLock()
        h = ctypes.c_void_p()
        check(library().mgx_create(index, ctypes.byref(h)))
        self.handle = h

    def close(self):
        if self.handle:
            library().mgx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- memory ----------------------------------------------------------------
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def upload(self, array, dtype=np.float32):
        host = np.ascontiguousarray(array, dtype=dtype)
        buf = DeviceBuffer(self, max(host.nbytes, 1))
        check(library().mgx_memcpy_h2d(self.handle, ctypes.c_void_p(buf.ptr),
                                       host.ctypes.data_as(ctypes.c_void_p), host.nbytes))
        return buf

    def download(self, buf, shape, dtype=np.float32):
        """Copy HBM -> numpy.  ``buf`` is a DeviceBuffer or a raw device address."""
        out = np.empty(shape, dtype=dtype)
        ptr = buf if isinstance(buf, int) else buf.ptr
        check(library().mgx_memcpy_d2h(self.handle, out.ctypes.data_as(ctypes.c_void_p),
                                       ctypes.c_void_p(ptr), out.nbytes))
        return out

    def synchronize(self):
        check(library().mgx_synchronize(self.handle))

    def timer_start(self):
        check(library().mgx_timer_start(self.handle))

    def timer_stop(self):
        ms = ctypes.c_float()
        check(library().mgx_timer_stop(self.handle, ctypes.byref(ms)))
        return ms.value

    # ---- per-stage device times of mgx_master (bench.py) ---------------------------
    def stage_timing(self, enable=True):
        check(library().mgx_stage_timing(self.handle, int(bool(enable))))

    def stage_times(self):
        """Milliseconds per stage of the last ``master`` call, keyed by stage name (``None`` = not run)."""
        ms = (ctypes.c_float * len(_native.STAGES))()
        check(library().mgx_stage_times(self.handle, ms))
        return {name: (float(v) if v >= 0 else None) for name, v in zip(_native.STAGES, ms)}

    # ---- the boundary ------------------------------------------------------------
    def master(self, target, n_target, reference, n_reference, native_config, result=None,
               result_no_limiter=None, result_no_limiter_normalized=None, want_report=True):
        """``mgx_master`` on device buffers.  Outputs are DeviceBuffers or None."""
        report = MgxReport() if want_report else None

        def p(b):
            return ctypes.c_void_p(b.ptr) if b is not None else None

        check(library().mgx_master(
            self.handle, p(target), n_target, p(reference), n_reference, ctypes.byref(native_config),
            p(result), p(result_no_limiter), p(result_no_limiter_normalized),
            ctypes.byref(report) if report is not None else None))
        return report


_default = {}


def default_device(index=0):
    """Process-wide Device for GPU ``index`` (created on first use)."""
    if index not in _default:
        _default[index] = Device(index)
    return _default[index]


def device_count():
    n = ctypes.c_int()
    rc = library().mgx_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0
