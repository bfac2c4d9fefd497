"""Audio file I/O for the host side of ``process`` (matchering/loader.py:30-74, saver.py:27-33,
results.py:25-46).

The reference delegates to ``soundfile`` (libsndfile) and falls back to an ``ffmpeg`` subprocess.
When ``soundfile`` is importable it is used here too (same formats, same behaviour); where it is
not -- as in the build image -- RIFF/WAVE files are read and written by the small numpy codec
below (PCM 8/16/24/32, IEEE float 32/64, plain and WAVE_FORMAT_EXTENSIBLE headers).  Scaling
follows libsndfile: integers are read as ``x / 2**(bits-1)`` and written as
``rint(x * (2**(bits-1) - 1))`` (clipped to the integer range instead of wrapping).
"""

import os
import struct
import subprocess

import numpy as np

from .log import Code, ModuleError, debug, info, warning
from .utils import random_file

try:                                    # pragma: no cover - not installed in the build image
    import soundfile as _sf
except Exception:                       # noqa: BLE001 - any import problem means "not available"
    _sf = None

WAV_SUBTYPES = ("PCM_U8", "PCM_16", "PCM_24", "PCM_32", "FLOAT", "DOUBLE")
_PCM, _FLOAT, _EXTENSIBLE = 1, 3, 0xFFFE


def check_format(extension, subtype=None):
    """``soundfile.check_format`` for the formats this build can write."""
    if _sf is not None:
        return _sf.check_format(extension, subtype)
    if extension.upper() not in ("WAV", "WAVE"):
        return False
    return subtype is None or subtype in WAV_SUBTYPES


# ---------------------------------------------------------------------------
# RIFF/WAVE codec
# ---------------------------------------------------------------------------
def read_wav(path):
    """(frames, channels) float64 in [-1, 1) and the sample rate."""
    with open(path, "rb") as fh:
        blob = fh.read()
    if len(blob) < 12 or blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise RuntimeError("Format not recognised: not a RIFF/WAVE file")
    fmt, data, pos = None, None, 12
    while pos + 8 <= len(blob):
        tag, size = blob[pos:pos + 4], struct.unpack("<I", blob[pos + 4:pos + 8])[0]
        body = blob[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            fmt = body
        elif tag == b"data":
            data = body
            if fmt is not None:
                break
        pos += 8 + size + (size & 1)
    if fmt is None or data is None or len(fmt) < 16:
        raise RuntimeError("Format not recognised: missing fmt or data chunk")
    code, channels, rate, _, block, bits = struct.unpack("<HHIIHH", fmt[:16])
    if code == _EXTENSIBLE and len(fmt) >= 26:
        code = struct.unpack("<H", fmt[24:26])[0]           # first two bytes of the sub-format GUID
    if channels < 1 or block < 1:
        raise RuntimeError("Format not recognised: bad channel count")
    frames = len(data) // block
    raw = np.frombuffer(data, dtype=np.uint8, count=frames * block)
    if code == _FLOAT and bits in (32, 64):
        out = raw.view("<f4" if bits == 32 else "<f8").astype(np.float64)
    elif code == _PCM and bits == 8:
        out = (raw.astype(np.float64) - 128.0) / 128.0
    elif code == _PCM and bits in (16, 32):
        out = raw.view("<i2" if bits == 16 else "<i4").astype(np.float64) / float(1 << (bits - 1))
    elif code == _PCM and bits == 24:
        b = raw.reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - (1 << 24), v)
        out = v.astype(np.float64) / float(1 << 23)
    else:
        raise RuntimeError(f"Format not recognised: WAVE format tag {code} with {bits} bits")
    return out.reshape(frames, channels), int(rate)


def write_wav(path, array, sample_rate, subtype):
    array = np.asarray(array, dtype=np.float64)
    if array.ndim == 1:
        array = array[:, None]
    frames, channels = array.shape
    if subtype in ("FLOAT", "DOUBLE"):
        code, bits = _FLOAT, 32 if subtype == "FLOAT" else 64
        payload = array.astype("<f4" if bits == 32 else "<f8").tobytes()
    elif subtype == "PCM_U8":
        code, bits = _PCM, 8
        payload = np.clip(np.rint(array * 127.0) + 128.0, 0, 255).astype(np.uint8).tobytes()
    elif subtype in ("PCM_16", "PCM_24", "PCM_32"):
        code, bits = _PCM, int(subtype[4:])
        top = float((1 << (bits - 1)) - 1)
        q = np.clip(np.rint(array * top), -top - 1.0, top).astype(np.int64)
        if bits == 24:
            u = (q & 0xFFFFFF).astype(np.uint32).reshape(-1)
            b = np.empty((u.size, 3), dtype=np.uint8)
            b[:, 0], b[:, 1], b[:, 2] = u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF
            payload = b.tobytes()
        else:
            payload = q.astype("<i2" if bits == 16 else "<i4").tobytes()
    else:
        raise TypeError(f"WAV format does not have {subtype} subtype")
    block = channels * bits // 8
    fmt = struct.pack("<HHIIHH", code, channels, int(sample_rate), int(sample_rate) * block, block, bits)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if code == _FLOAT:
        chunks += b"fact" + struct.pack("<II", 4, frames)
    chunks += b"data" + struct.pack("<I", len(payload)) + payload + (b"\x00" if len(payload) & 1 else b"")
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)


def _read(path):
    if _sf is not None:
        return _sf.read(path, always_2d=True)
    return read_wav(path)


# ---------------------------------------------------------------------------
# matchering.loader.load / matchering.saver.save
# ---------------------------------------------------------------------------
def load(file: str, file_type: str, temp_folder: str):
    """loader.py:30-47: returns ``(sound (n, channels) float64, sample_rate)``; raises
    ``ModuleError(4001 | 4101)`` when the file cannot be decoded (after trying ffmpeg)."""
    file_type = file_type.upper()
    sound, sample_rate = None, None
    debug(f"Loading the {file_type} file: '{file}'...")
    try:
        sound, sample_rate = _read(file)
    except (RuntimeError, OSError) as e:
        debug(e)
        if "unknown format" in str(e) or "Format not recognised" in str(e):
            sound, sample_rate = _load_with_ffmpeg(file, file_type, temp_folder)
    if sound is None or sample_rate is None:
        raise ModuleError(Code.ERROR_TARGET_LOADING if file_type == "TARGET" else Code.ERROR_REFERENCE_LOADING)
    debug(f"The {file_type} file is loaded")
    return sound, sample_rate


def _load_with_ffmpeg(file, file_type, temp_folder):
    """loader.py:50-74: decode through an ``ffmpeg`` subprocess into a temporary WAV."""
    sound, sample_rate = None, None
    debug(f"Trying to load '{file}' with ffmpeg...")
    temp_file = os.path.join(temp_folder, random_file(prefix="temp"))
    with open(os.devnull, "w") as devnull:
        try:
            subprocess.check_call(["ffmpeg", "-i", file, temp_file], stdout=devnull, stderr=devnull)
            sound, sample_rate = _read(temp_file)
            if file_type == "TARGET":
                warning(Code.WARNING_TARGET_IS_LOSSY)
            else:
                info(Code.INFO_REFERENCE_IS_LOSSY)
            os.remove(temp_file)
        except FileNotFoundError:
            debug("ffmpeg is not found in the system! "
                  "Download, install and add it to PATH: https://www.ffmpeg.org/download.html")
        except subprocess.CalledProcessError:
            debug(f"ffmpeg cannot convert '{file}' to .wav!")
    return sound, sample_rate


def save(file: str, result: np.ndarray, sample_rate: int, subtype: str, name: str = "result") -> None:
    """saver.py:27-33."""
    debug(f"Saving the {name.upper()} {sample_rate} Hz Stereo {subtype} to: '{file}'...")
    if _sf is not None:
        _sf.write(file, result, sample_rate, subtype)
    else:
        write_wav(file, result, sample_rate, subtype)
    debug(f"'{file}' is saved")
