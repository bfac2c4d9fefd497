"""A/B previews (matchering/preview_creator.py:30-94): the loudest ``preview_size`` window of the
RESULT (RMS over both channels, windows every ``preview_analysis_step``), cut from target and
result alike, faded in and out, saved.

Two paths.  ``PreviewRequest`` rides along with ``stages.main``: the window energies are one segmented sum
over the result while it is still in HBM (``mgx_window_energy``), the two pieces are cut, clipped and
faded there (``mgx_preview_cut``) and only they -- 30 s each, quantised on the device when the preview
file is integer PCM -- cross PCIe.  ``create_preview`` is the same thing in numpy on host arrays, for
callers that hold the arrays already (and for the CPU tests)."""

import numpy as np

from .audio_io import save
from .config import Config
from .log import Code, debug, debug_line, info
from .results import Result
from .utils import time_str


def _window_starts(length, size, step):
    if size > length:
        return np.array([0]), length                      # dsp.py:131-132: the whole array is the only window
    return np.arange((length - size) // step + 1) * step, size


def _loudest_window(result, size, step):
    """argmax over windows of the mean square of all samples in the window (dsp.py:128-143)."""
    starts, size = _window_starts(result.shape[0], size, step)
    csum = np.concatenate(([0.0], np.cumsum(np.einsum("ij,ij->i", result, result))))
    energy = csum[starts + size] - csum[starts]
    return int(np.argmax(energy)), starts, size


def _fade(array, fade_size):
    """dsp.py:146-152: linear fade-in and fade-out of ``fade_size`` frames."""
    array = np.array(array, dtype=np.float64, copy=True)
    ramp = np.linspace(0, 1, fade_size)
    array[:fade_size] *= ramp[:, None]
    array[array.shape[0] - fade_size:] *= ramp[::-1, None]
    return array


class PreviewRequest:
    """What ``stages.main(..., preview=request)`` needs to cut the previews on the device, and where it
    leaves them: ``begin`` (first frame of the loudest window), ``target_piece`` / ``result_piece`` (host
    arrays: float32 (m, 2), or the integer PCM of ``encodings`` -- PCM_16 / PCM_24 / PCM_32 or None each)."""

    def __init__(self, config: Config, preview_target: Result = None, preview_result: Result = None, encodings=(None, None)):
        self.size = int(config.preview_size)
        self.step = int(config.preview_analysis_step)
        self.fade_size = int(config.preview_fade_size)
        self.fade_coefficient = int(config.preview_fade_coefficient)
        self.threshold = float(config.threshold)
        self.want_target, self.want_result = preview_target is not None, preview_result is not None
        self.encodings = tuple(encodings)
        self.begin = self.frames = self.fade = None
        self.target_piece = self.result_piece = None

    def plan(self, energy, frames):
        """Window choice and fade length from the window energies (preview_creator.py:52-78)."""
        index = int(np.argmax(energy))                                   # first of equals, like numpy.argmax there
        self.begin = index * self.step if self.size <= frames else 0
        self.frames = min(self.size, frames)
        whole = self.frames == frames                                    # preview_creator.py:69: no fades then
        self.fade = 0 if whole else int(min(self.fade_size, self.frames // self.fade_coefficient))
        return self.begin, self.frames, self.fade


def save_previews(request: PreviewRequest, config: Config, preview_target: Result, preview_result: Result) -> None:
    """The saving half of preview_creator.py:30-94 for pieces that ``stages.main`` cut on the device."""
    debug_line()
    info(Code.INFO_MAKING_PREVIEWS)
    debug(f"previews: up to {request.size / config.internal_sample_rate} s, searched in steps of "
          f"{request.step / config.internal_sample_rate} s")
    debug(f"loudest window of the result: {time_str(request.begin, config.internal_sample_rate)} to "
          f"{time_str(request.begin + request.frames, config.internal_sample_rate)}")
    if preview_target:
        save(preview_target.file, request.target_piece, config.internal_sample_rate, preview_target.subtype, "target preview")
    if preview_result:
        save(preview_result.file, request.result_piece, config.internal_sample_rate, preview_result.subtype, "result preview")


def create_preview(target: np.ndarray, result: np.ndarray, config: Config, preview_target: Result,
                   preview_result: Result) -> None:
    debug_line()
    info(Code.INFO_MAKING_PREVIEWS)
    target = np.clip(target, -config.threshold, config.threshold)            # dsp.py:109-110
    size, step = int(config.preview_size), int(config.preview_analysis_step)
    debug(f"previews: up to {size / config.internal_sample_rate} s, searched in steps of "
          f"{step / config.internal_sample_rate} s")
    index, starts, size = _loudest_window(np.asarray(result, dtype=np.float64), size, step)
    begin = int(starts[index])
    target_piece = np.array(target[begin:begin + size], dtype=np.float64)
    result_piece = np.array(result[begin:begin + size], dtype=np.float64)
    debug(f"loudest window of the result: {time_str(begin, config.internal_sample_rate)} to "
          f"{time_str(begin + result_piece.shape[0], config.internal_sample_rate)}")
    if result.shape[0] != result_piece.shape[0]:
        fade_size = int(min(config.preview_fade_size, result_piece.shape[0] // config.preview_fade_coefficient))
        target_piece, result_piece = _fade(target_piece, fade_size), _fade(result_piece, fade_size)
    if preview_target:
        save(preview_target.file, target_piece, config.internal_sample_rate, preview_target.subtype, "target preview")
    if preview_result:
        save(preview_result.file, result_piece, config.internal_sample_rate, preview_result.subtype, "result preview")
