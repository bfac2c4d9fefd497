"""Device-side plumbing for the Python host: one ``Device`` = one ``mgx_handle``
(one MI355X, one HIP stream).  Buffers are plain HBM allocations addressed by
integer pointers; numpy arrays cross the PCIe boundary only in ``upload`` /
``download``.  No PyTorch: the north star keeps the host a thin ctypes caller.
"""

import bisect
import ctypes
import threading
import weakref
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _native
from ._native import MgxReport, check, library


class _PinnedPool:
    """Page-locked host blocks (``mgx_host_alloc``), recycled by size class.

    ``empty(shape, dtype)`` returns a numpy array living in such a block; when the array (and every
    view of it) is garbage collected the block goes back to the free list instead of to the OS --
    page-locking 170 MB costs milliseconds, about as much as the copy it is meant to speed up.  Arrays
    handed out here are recognised by ``Device.upload`` / ``download`` (``holds``), which then move them
    with asynchronous DMA instead of a staged copy."""

    KEEP = 8                                   # free blocks kept per size class
    MAX_FREE_BYTES = 4 << 30                   # ... and in total: above 16 MiB every track length is its own size
                                               # class, so a long batch of varied tracks would otherwise pile up
                                               # page-locked memory without bound

    def __init__(self):
        self.lock = threading.Lock()
        self.free = {}                         # size class -> [address, ...]
        self.free_bytes = 0
        self.order = []                        # (size, address) of the free blocks, oldest first
        self.starts, self.ends = [], []        # live blocks, sorted by address

    def empty(self, shape, dtype=np.float32):
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        size = _size_class(max(nbytes, 1))
        with self.lock:
            stack = self.free.get(size)
            address = stack.pop() if stack else None
            if address is not None:
                self.free_bytes -= size
                self.order.remove((size, address))
        if address is None:
            ptr = ctypes.c_void_p()
            check(library().mgx_host_alloc(size, ctypes.byref(ptr)))
            address = ptr.value
            with self.lock:
                i = bisect.bisect_left(self.starts, address)
                self.starts.insert(i, address)
                self.ends.insert(i, address + size)
        raw = (ctypes.c_char * size).from_address(address)
        array = np.frombuffer(raw, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        weakref.finalize(raw, self._give_back, address, size)
        return array

    def _give_back(self, address, size):
        doomed = []
        with self.lock:
            stack = self.free.setdefault(size, [])
            if len(stack) < self.KEEP and size <= self.MAX_FREE_BYTES:
                stack.append(address)
                self.order.append((size, address))
                self.free_bytes += size
                while self.free_bytes > self.MAX_FREE_BYTES:          # trim the blocks that have waited longest
                    old_size, old_address = self.order.pop(0)
                    self.free[old_size].remove(old_address)
                    self.free_bytes -= old_size
                    doomed.append(old_address)
            else:
                doomed.append(address)
            for a in doomed:
                i = bisect.bisect_left(self.starts, a)
                del self.starts[i], self.ends[i]
        for a in doomed:
            try:
                library().mgx_host_free(ctypes.c_void_p(a))
            except Exception:
                pass

    def trim(self):
        """Give every free block back to the OS (``process_batch`` calls this when it is done)."""
        with self.lock:
            doomed = [a for _, a in self.order]
            self.order, self.free, self.free_bytes = [], {}, 0
            for a in doomed:
                i = bisect.bisect_left(self.starts, a)
                del self.starts[i], self.ends[i]
        for a in doomed:
            try:
                library().mgx_host_free(ctypes.c_void_p(a))
            except Exception:
                pass

    def holds(self, array):
        """True when ``array``'s bytes lie inside one of this pool's blocks."""
        first = array.ctypes.data
        with self.lock:
            i = bisect.bisect_right(self.starts, first) - 1
            return i >= 0 and first + array.nbytes <= self.ends[i]


pinned = _PinnedPool()
_copiers = ThreadPoolExecutor(max_workers=4, thread_name_prefix="mgx-stage")
_STAGE_CHUNK = 8 << 20


def _size_class(nbytes):
    """Allocation sizes: powers of two up to 16 MiB, multiples of 16 MiB above."""
    size = 1 << 16
    while size < nbytes:
        size <<= 1
    return size if size <= (1 << 24) else -(-nbytes // (1 << 24)) * (1 << 24)


def pcm_bits(array):
    """Width of the integer PCM samples an array holds (16, 24, 32), 0 for a float array."""
    if array.dtype == np.int16:
        return 16
    if array.dtype == np.int32:
        return 32
    if array.dtype == np.uint8:
        return 24
    return 0


class DeviceFrames:
    """(frames, 2) float32 audio resident in HBM: what ``stages.main`` accepts in place of a numpy array when
    the caller has uploaded (and looked at) a track already -- ``core.process`` does, for the checks."""

    def __init__(self, buf, frames):
        self.buf, self.frames = buf, int(frames)

    @property
    def shape(self):
        return (self.frames, 2)

    def release(self):
        self.buf.release()


class DeviceBuffer:
    """An HBM allocation owned by a Device.  ``release`` (or garbage collection) hands the block back to
    the device's free list -- hipMalloc / hipFree of a 170 MB block cost milliseconds and hipFree waits
    for the whole GPU, which would serialise the lanes of a batch -- and ``Device.close`` frees the lot."""

    def __init__(self, device, nbytes):
        self.device = device
        self.nbytes = int(nbytes)
        self.capacity = _size_class(max(self.nbytes, 1))
        self.ptr = device._take_block(self.capacity)

    def release(self):
        if self.ptr and self.device.handle:
            self.device._give_block(self.capacity, self.ptr)
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
        self._keep_until_sync = []            # host arrays with a queued DMA still reading them
        self._blocks = {}                     # free HBM blocks by size class
        self._free_hbm = 0                    # ... and their total size
        h = ctypes.c_void_p()
        check(library().mgx_create(index, ctypes.byref(h)))
        self.handle = h

    def close(self):
        if self.handle:
            for stack in self._blocks.values():
                for ptr in stack:
                    library().mgx_free(self.handle, ctypes.c_void_p(ptr))
            self._blocks.clear()
            library().mgx_destroy(self.handle)
            self.handle = None

    # HBM blocks are recycled: a block released while kernels of THIS handle may still read it is only
    # reused by later work on the same stream, which is ordered behind them
    KEEP_BLOCKS = 6
    MAX_FREE_HBM = 24 << 30                  # free HBM blocks kept per handle, in total (a lane's share of 288 GB)

    def _take_block(self, capacity):
        with self.lock:
            stack = self._blocks.get(capacity)
            if stack:
                self._free_hbm -= capacity
                return stack.pop()
        ptr = ctypes.c_void_p()
        check(library().mgx_malloc(self.handle, capacity, ctypes.byref(ptr)))
        return ptr.value

    def _give_block(self, capacity, ptr):
        with self.lock:
            stack = self._blocks.setdefault(capacity, [])
            if len(stack) < self.KEEP_BLOCKS and self._free_hbm + capacity <= self.MAX_FREE_HBM:
                stack.append(ptr)
                self._free_hbm += capacity
                return
        library().mgx_free(self.handle, ctypes.c_void_p(ptr))

    def trim(self):
        """Free every recycled HBM block of this handle."""
        with self.lock:
            stacks, self._blocks, self._free_hbm = self._blocks, {}, 0
        for stack in stacks.values():
            for ptr in stack:
                library().mgx_free(self.handle, ctypes.c_void_p(ptr))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- memory ----------------------------------------------------------------
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def upload(self, array, dtype=np.float32):
        """numpy -> a new HBM buffer.  Returns as soon as the copy is queued on the stream: arrays from
        ``pinned.empty`` go by DMA straight from where they are; any other array is moved into a
        pinned block by four host threads, chunk by chunk, each chunk's DMA queued as it lands (about
        twice the rate of a pageable hipMemcpy).  The source must not be written before ``synchronize``
        or the next blocking call on this device.  ``dtype=None`` keeps the array's own type."""
        host = np.ascontiguousarray(array) if dtype is None else np.ascontiguousarray(array, dtype=dtype)
        buf = DeviceBuffer(self, max(host.nbytes, 1))
        if host.nbytes == 0:
            return buf
        lib = library()
        if pinned.holds(host):
            check(lib.mgx_memcpy_h2d_async(self.handle, ctypes.c_void_p(buf.ptr),
                                           host.ctypes.data_as(ctypes.c_void_p), host.nbytes))
            self._keep_until_sync.append(host)
            return buf
        stage = pinned.empty((host.nbytes,), np.uint8)
        src, dst = host.ctypes.data, stage.ctypes.data
        pieces = [(o, min(_STAGE_CHUNK, host.nbytes - o)) for o in range(0, host.nbytes, _STAGE_CHUNK)]
        landed = [_copiers.submit(ctypes.memmove, dst + o, src + o, n) for o, n in pieces]
        for (o, n), done in zip(pieces, landed):
            done.result()
            check(lib.mgx_memcpy_h2d_async(self.handle, ctypes.c_void_p(buf.ptr + o), ctypes.c_void_p(dst + o), n))
        self._keep_until_sync.append(stage)
        return buf

    def download(self, buf, shape, dtype=np.float32, wait=True):
        """HBM -> a numpy array in pinned memory (``pinned.empty``).  ``buf`` is a DeviceBuffer or a raw
        device address.  With ``wait=False`` the copy is only queued: call ``synchronize`` before
        reading the array."""
        out = pinned.empty(shape, dtype)
        ptr = buf if isinstance(buf, int) else buf.ptr
        if out.nbytes:
            check(library().mgx_memcpy_d2h_async(self.handle, out.ctypes.data_as(ctypes.c_void_p),
                                                 ctypes.c_void_p(ptr), out.nbytes))
        if wait:
            self.synchronize()
        return out

    # ---- integer PCM at the boundary (mgx_pcm_decode / mgx_pcm_encode) ---------------------------
    def upload_frames(self, array):
        """(n, channels) audio -> float32 frames in HBM.  Float arrays are uploaded as float32; integer
        arrays are what a PCM file holds (``audio_io.PCM_DTYPES``: int16, int32 with the sample in the high
        bits, or uint8 (n, channels * 3) of packed 24-bit samples): they cross PCIe as they are and become
        ``v / 2**(bits - 1)`` on the device."""
        bits = pcm_bits(array)
        if bits == 0:
            return self.upload(array)
        raw = self.upload(array, dtype=None)
        samples = array.size // 3 if bits == 24 else array.size
        out = DeviceBuffer(self, max(samples * 4, 1))
        check(library().mgx_pcm_decode(self.handle, ctypes.c_void_p(raw.ptr), samples, bits, ctypes.c_void_p(out.ptr)))
        raw.release()           # (recycled by later work on this stream only, which is ordered behind the decode)
        return out

    def peak_count(self, buf, samples):
        """dsp.py:49-54 count_max_peaks of float32 samples in HBM: (largest magnitude, samples on it)."""
        peak, count = ctypes.c_double(), ctypes.c_int64()
        ptr = buf.ptr if hasattr(buf, "ptr") else buf.buf.ptr
        check(library().mgx_peak_count(self.handle, ctypes.c_void_p(ptr), int(samples), ctypes.byref(peak),
                                       ctypes.byref(count)))
        return peak.value, count.value

    def download_pcm(self, buf, frames, channels, bits, wait=True):
        """float32 frames in HBM -> integer PCM on the host (pinned): int16 / int32 (n, channels), or uint8
        (n, channels * 3) for packed 24-bit.  Quantised on the device as libsndfile would on the host."""
        samples = frames * channels
        nbytes = samples * bits // 8
        pcm = DeviceBuffer(self, max(nbytes, 1))
        check(library().mgx_pcm_encode(self.handle, ctypes.c_void_p(buf.ptr), samples, bits, ctypes.c_void_p(pcm.ptr)))
        shape, dtype = {16: ((frames, channels), np.int16), 32: ((frames, channels), np.int32),
                        24: ((frames, channels * 3), np.uint8)}[bits]
        out = self.download(pcm, shape, dtype, wait=False)
        pcm.release()
        if wait:
            self.synchronize()
        return out

    # ---- previews on frames that are still in HBM (mgx_window_energy / mgx_preview_cut) ----------
    def window_energy(self, buf, frames, size, step):
        """dsp.py:128-143: sum of squares (both channels) of every window of ``size`` frames every ``step``."""
        size = min(int(size), int(frames))
        count = (int(frames) - size) // int(step) + 1
        energy = (ctypes.c_double * count)()
        got = ctypes.c_int64()
        check(library().mgx_window_energy(self.handle, ctypes.c_void_p(buf.ptr), int(frames), size, int(step), energy,
                                          count, ctypes.byref(got)))
        return np.frombuffer(energy, dtype=np.float64, count=got.value).copy()

    def preview_cut(self, buf, frames, begin, size, fade, clip_limit=0.0):
        """Frames [begin, begin + size) clipped (``clip_limit`` > 0) and faded: a new DeviceBuffer."""
        out = DeviceBuffer(self, max(int(size) * 8, 1))
        check(library().mgx_preview_cut(self.handle, ctypes.c_void_p(buf.ptr), int(frames), int(begin), int(size),
                                        int(fade), float(clip_limit), ctypes.c_void_p(out.ptr)))
        return out

    def synchronize(self):
        check(library().mgx_synchronize(self.handle))
        self._keep_until_sync.clear()

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
               result_no_limiter=None, result_no_limiter_normalized=None, want_report=True, fir=None):
        """``mgx_master`` on device buffers.  Outputs are DeviceBuffers or None.  ``fir`` (a DeviceBuffer
        holding [2][fft_size] float32) replaces the designed matching FIR: ``mgx_master_with_fir``."""
        report = MgxReport() if want_report else None

        def p(b):
            return ctypes.c_void_p(b.ptr) if b is not None else None

        rep = ctypes.byref(report) if report is not None else None
        if fir is None:
            check(library().mgx_master(
                self.handle, p(target), n_target, p(reference), n_reference, ctypes.byref(native_config),
                p(result), p(result_no_limiter), p(result_no_limiter_normalized), rep))
        else:
            check(library().mgx_master_with_fir(
                self.handle, p(target), n_target, p(reference), n_reference, ctypes.byref(native_config), p(fir),
                p(result), p(result_no_limiter), p(result_no_limiter_normalized), rep))
        return report

    def last_fir(self):
        """(device address, taps) of the FIR pair the last ``master`` designed or was given."""
        ptr, taps = ctypes.c_void_p(), ctypes.c_int32()
        check(library().mgx_last_fir(self.handle, ctypes.byref(ptr), ctypes.byref(taps)))
        return ptr.value, taps.value

    # ---- RCCL: the FIR exchange of the multi-GPU path --------------------------------
    def comm_init(self, rank, world, exchange=None):
        """Join a communicator of ``world`` ranks.  ``exchange(payload, size) -> bytes`` must hand rank 0's
        128-byte id to every rank (bench.Ranks.broadcast_bytes does); not needed for world == 1."""
        from .ranks import single_node_rccl_defaults

        single_node_rccl_defaults()          # (only what the host has not set; nothing when the ranks span machines)
        lib = library()
        uid = ctypes.create_string_buffer(128)
        if rank == 0:
            check(lib.mgx_comm_unique_id(uid))
        payload = uid.raw if exchange is None else exchange(uid.raw, 128)
        check(lib.mgx_comm_init(self.handle, ctypes.c_char_p(payload), rank, world))

    def comm_broadcast(self, buf, count, root=0):
        ptr = buf if isinstance(buf, int) else buf.ptr
        check(library().mgx_comm_broadcast_f32(self.handle, ctypes.c_void_p(ptr), count, root))

    def comm_destroy(self):
        check(library().mgx_comm_destroy(self.handle))


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


def pci_bus_id(index=0):
    """PCI address of GPU ``index`` ("0000:0d:00.0"): which physical device a rank or lane sits on."""
    buf = ctypes.create_string_buffer(32)
    check(library().mgx_device_pci_bus_id(int(index), buf, 32))
    return buf.value.decode("ascii", "replace")
