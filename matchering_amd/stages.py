"""``main``: the drop-in for matchering/stages.py:210-272.

Same call signature and return convention as the reference -- two (n, 2) arrays
in, a triple ``(result, result_no_limiter, result_no_limiter_normalized)`` out
with ``None`` for outputs that were not requested -- but the four stages run as
HIP kernels on an MI355X through ``mgx_master`` (include/mgx.h).  Arrays come
back as float32 (n, 2) C-ordered; the reference returns float64.  The progress
codes 2004-2007 are emitted in the reference's order (stages.py:52,117,147,182)
and the per-stage scalars it prints through ``debug`` are reported from the
device-side values.
"""

import numpy as np

from .config import Config
from .device import DeviceFrames, default_device
from .log import Code, debug, debug_line, info
from .utils import to_db


PCM_BITS = {"PCM_16": 16, "PCM_24": 24, "PCM_32": 32}


def _as_frames(array, name):
    """(n, 2) frames as they will cross PCIe: float32, or the integer PCM of a file as it is (int16 / int32,
    or packed 24-bit uint8 (n, 6), audio_io): those are decoded on the device."""
    if isinstance(array, DeviceFrames):                  # uploaded by the caller already
        return array
    array = np.asarray(array)
    if array.dtype == np.uint8:                          # packed 24-bit PCM
        if array.ndim != 2 or array.shape[1] != 6:
            raise ValueError(f"{name} must hold (n, 2) packed 24-bit frames, got {array.shape}")
        return np.ascontiguousarray(array)
    if array.ndim != 2 or array.shape[1] != 2:
        raise ValueError(f"{name} must have shape (n, 2), got {array.shape}")
    if array.dtype in (np.int16, np.int32):
        return np.ascontiguousarray(array)
    return np.ascontiguousarray(array, dtype=np.float32)


def main(target: np.ndarray, reference: np.ndarray, config: Config, need_default: bool = True,
         need_no_limiter: bool = False, need_no_limiter_normalized: bool = False, device=None, fir=None,
         encodings=None, preview=None):
    # (``device``: the handle to run on, default the process-wide one; ``fir``: a DeviceBuffer with a
    # matching FIR to apply instead of designing one -- batch.master_album; ``encodings``: per output,
    # None for float32 frames or "PCM_16" / "PCM_24" / "PCM_32" for the integer samples a file of that
    # subtype holds, quantised on the device (saver.py:27-33 does it on the host) -- int16 / int32 (n, 2),
    # or uint8 (n, 6) for packed 24-bit.  All additions to the reference's signature, keyword-only in
    # spirit.  ``target`` / ``reference`` may be int16 or int32 PCM as read from a file.  ``preview``: a
    # preview.PreviewRequest -- the A/B previews of preview_creator.py:30-94 are cut from the first requested
    # output and from the target while both are still in HBM, and left in the request.)
    dev = device if device is not None else default_device()
    target = _as_frames(target, "target")
    reference = _as_frames(reference, "reference")
    n, nr = target.shape[0], reference.shape[0]
    native = config.to_native()

    debug_line()
    info(Code.INFO_MATCHING_LEVELS)
    debug(f"analysis pieces: at most {config.max_piece_size} frames "
          f"({config.max_piece_size / config.internal_sample_rate:.2f} s) each")
    with dev.lock:
        t_dev = target.buf if isinstance(target, DeviceFrames) else dev.upload_frames(target)
        r_dev = reference.buf if isinstance(reference, DeviceFrames) else dev.upload_frames(reference)
        outs = [dev.alloc(n * 8) if need else None
                for need in (need_default, need_no_limiter, need_no_limiter_normalized)]
        try:
            report = dev.master(t_dev, n, r_dev, nr, native, *outs, fir=fir)
            debug(f"target: {report.target_divisions} pieces of {report.target_piece} frames, "
                  f"{report.target_loud_count} of them loud; reference: {report.reference_divisions} pieces of "
                  f"{report.reference_piece} frames, {report.reference_loud_count} loud")
            if not np.isclose(report.final_amplitude_coefficient, 1.0):
                debug("the reference peaks below the threshold: it was scaled up for matching and the result "
                      f"is scaled back by {to_db(report.final_amplitude_coefficient)}")
            debug(f"level match: {to_db(report.rms_coefficient)} on the target")
            debug_line()
            info(Code.INFO_MATCHING_FREQS)
            debug_line()
            info(Code.INFO_CORRECTING_LEVELS)
            kept = len(report.correction_coefficients)               # (mgx_report keeps the first 16 coefficients)
            for step in range(min(config.rms_correction_steps, kept)):
                debug(f"correction round {step + 1}: {to_db(report.correction_coefficients[step])}")
            if config.rms_correction_steps > kept:
                debug(f"... and {config.rms_correction_steps - kept} more rounds")
            debug_line()
            info(Code.INFO_FINALIZING)
            if need_no_limiter_normalized:
                debug(f"unlimited result normalised by {to_db(report.normalize_coefficient)} to reach the threshold")
            if need_default and not report.limiter_active:
                debug("the result stays under the threshold: the limiter passes it through")
            # queued one behind the other, then ONE wait; the arrays live in pinned host memory
            formats = encodings if encodings is not None else (None, None, None)
            pieces = []
            if preview is not None:
                mastered = next(b for b in outs if b is not None)        # core.py:111: the first rendering there is
                begin, size, fade = preview.plan(dev.window_energy(mastered, n, preview.size, preview.step), n)
                for want, src, limit, fmt in ((preview.want_target, t_dev, preview.threshold, preview.encodings[0]),
                                              (preview.want_result, mastered, 0.0, preview.encodings[1])):
                    piece = dev.preview_cut(src, n, begin, size, fade, limit) if want else None
                    pieces.append(piece)
                    host = (None if piece is None else dev.download(piece, (size, 2), wait=False) if fmt is None
                            else dev.download_pcm(piece, size, 2, PCM_BITS[fmt], wait=False))
                    if src is t_dev:
                        preview.target_piece = host
                    else:
                        preview.result_piece = host
            results = tuple(None if b is None
                            else dev.download(b, (n, 2), wait=False) if fmt is None
                            else dev.download_pcm(b, n, 2, PCM_BITS[fmt], wait=False)
                            for b, fmt in zip(outs, formats))
            dev.synchronize()
            for piece in pieces:
                if piece is not None:
                    piece.release()
        finally:
            for b in (t_dev, r_dev, *outs):
                if b is not None:
                    b.release()
    return results
