"""``process``: the public entry point with the reference's signature, log codes and exceptions
(matchering/core.py:32-121).  The work is split the way this package needs it: two host-side steps
around the one call that runs on the GPU.

    files --_read_pair--> frames at the internal rate --stages.main (MI355X)--> three renderings
          --_write_results--> files (+ optional previews)

Integer PCM and float32 WAVE files are mapped, not decoded: their samples go to the GPU as the file holds
them (``mgx_pcm_decode``), the target's peak statistics are taken there (``mgx_peak_count``), and renderings
come back quantised for the ``Result`` files' subtype (``mgx_pcm_encode``) -- see ``_read_pair`` and
``_wanted_encodings``.
"""

import os

import numpy as np

from .audio_io import load, pcm_channels, pcm_to_float, save
from .checker import check, check_equality
from .config import Config
from .log import Code, ModuleError, debug, debug_line, info
from .preview import PreviewRequest, create_preview, save_previews
from .results import Result
from .stages import main
from .utils import get_temp_folder


def _wanted_renderings(results):
    """Which of stages.main's three outputs the requested files need (core.py:77-86)."""
    limited = plain = normalized = False
    for item in results:
        if item.use_limiter:
            limited = True
        elif item.normalize:
            normalized = True
        else:
            plain = True
    return limited, plain, normalized


def _wanted_encodings(results):
    """Per rendering, the PCM subtype the GPU can quantise to directly: every file made from that rendering
    is a WAVE file of one and the same integer subtype (saver.py:27-33 would quantise on the host)."""
    from .stages import PCM_BITS

    wanted = [set(), set(), set()]
    for item in results:
        slot = 0 if item.use_limiter else (2 if item.normalize else 1)
        wave = os.path.splitext(item.file)[1][1:].upper() in ("WAV", "WAVE")
        wanted[slot].add(item.subtype if wave and item.subtype in PCM_BITS else None)
    return tuple(next(iter(w)) if len(w) == 1 else None for w in wanted)


def _file_encoding(item):
    """The integer subtype of one WAVE file (quantised on the GPU), or None (float frames, host codec)."""
    from .stages import PCM_BITS

    if item is None or os.path.splitext(item.file)[1][1:].upper() not in ("WAV", "WAVE"):
        return None
    return item.subtype if item.subtype in PCM_BITS else None


def _gpu():
    """The default device, or None where there is none (the CPU tests put a stand-in behind ``main``)."""
    try:
        from .device import default_device

        return default_device()
    except Exception:       # noqa: BLE001 -- no library, no GPU: the host-side checks do all of it
        return None


def _read_pair(target_path, reference_path, config, temp_folder):
    """Load and check both tracks (core.py:52-74); raises ModuleError with the reference's codes.  Integer
    PCM and float32 tracks that need no channel or rate conversion go to the GPU at once: they are decoded
    there, the target's peak statistics (checker.py:118-130) are taken there, and ``main`` receives them
    resident."""
    from .device import DeviceFrames

    dev = _gpu()
    tracks, resident = [], []
    for path, role in ((target_path, "target"), (reference_path, "reference")):
        audio, rate = load(path, role, temp_folder, pcm=True)    # 16/24/32-bit WAVE: decoded on the GPU
        peaks, frames = None, None
        # (integer PCM, or float32 frames as a FLOAT file holds them)
        direct = (dev is not None and (audio.dtype.kind in "iu" or audio.dtype == np.float32)
                  and pcm_channels(audio) == 2 and rate == config.internal_sample_rate and audio.shape[0] > 0)
        if direct:
            with dev.lock:
                frames = DeviceFrames(dev.upload_frames(audio), audio.shape[0])
                if role == "target":
                    peaks = dev.peak_count(frames, 2 * audio.shape[0])
        try:
            tracks.append(check(audio, rate, config, role, peaks=peaks))
        except Exception:
            for f in resident + [frames]:
                if f is not None:
                    f.release()
            raise
        resident.append(frames)
    (target, target_rate), (reference, reference_rate) = tracks
    try:
        if not config.allow_equality:
            check_equality(target, reference)
        consistent = (
            target_rate == reference_rate == config.internal_sample_rate
            and pcm_channels(target) == pcm_channels(reference) == 2
            and min(target.shape[0], reference.shape[0]) > config.fft_size
        )
        if not consistent:
            raise ModuleError(Code.ERROR_VALIDATION)
    except Exception:
        for f in resident:
            if f is not None:
                f.release()
        raise
    return target, reference, resident


def _same_file(a, b):
    try:
        return os.path.exists(a) and os.path.samefile(a, b)
    except OSError:
        return False


def _write_results(results, renderings, sample_rate):
    """One file per Result, each from the rendering it asked for (core.py:95-108)."""
    limited, plain, normalized = renderings
    for item in results:
        audio = limited if item.use_limiter else (normalized if item.normalize else plain)
        save(item.file, audio, sample_rate, item.subtype)


def process(target: str, reference: str, results: list, config: Config = None,
            preview_target: Result = None, preview_result: Result = None):
    config = Config() if config is None else config
    debug("matchering_amd: the MI355X path behind the API of https://github.com/sergree/matchering")
    debug_line()
    info(Code.INFO_LOADING)
    if not results:
        raise RuntimeError("The result list is empty")
    temp_folder = config.temp_folder or get_temp_folder(results)

    target_audio, reference_audio, resident = _read_pair(target, reference, config, temp_folder)
    previews = bool(preview_target or preview_result)
    # With a GPU the previews are cut on it from the frames stages.main leaves in HBM (preview.PreviewRequest):
    # only the two 30 s pieces cross PCIe, and the renderings keep their integer encodings.  Without one (the
    # CPU tests put a stand-in behind ``main``) they are cut from float renderings on the host.
    request = None
    if previews and _gpu() is not None:
        request = PreviewRequest(config, preview_target, preview_result,
                                 (_file_encoding(preview_target), _file_encoding(preview_result)))
    encodings = None if (previews and request is None) else _wanted_encodings(results)
    extra = {"preview": request} if request is not None else {}
    renderings = main(resident[0] if resident[0] is not None else target_audio,
                      resident[1] if resident[1] is not None else reference_audio,
                      config, *_wanted_renderings(results), encodings=encodings, **extra)   # (releases the resident frames)
    del reference_audio

    debug_line()
    info(Code.INFO_EXPORTING)
    if previews and request is None and any(_same_file(item.file, target) for item in results):
        # the target may be a read-only mapping of its file (audio_io.read_wav): a result written over that
        # file would pull the mapping from under the preview cut below
        target_audio = np.array(target_audio, copy=True)
    _write_results(results, renderings, config.internal_sample_rate)

    if request is not None:
        save_previews(request, config, preview_target, preview_result)
    elif previews:
        mastered = next(audio for audio in renderings if audio is not None)
        create_preview(pcm_to_float(target_audio), mastered, config, preview_target, preview_result)

    debug_line()
    info(Code.INFO_COMPLETED)
