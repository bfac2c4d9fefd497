"""Audio file I/O for the host side of ``process`` (matchering/loader.py:30-74, saver.py:27-33,
results.py:25-46).

The reference delegates to ``soundfile`` (libsndfile) and falls back to an ``ffmpeg`` subprocess.
When ``soundfile`` is importable it is used here too (same formats, same behaviour); where it is
not -- as in the build image -- RIFF/WAVE and AIFF / AIFF-C files are read and written by the small
numpy codecs below (WAVE: PCM 8/16/24/32, IEEE float 32/64, plain and WAVE_FORMAT_EXTENSIBLE headers;
AIFF: big-endian PCM 8/16/24/32, AIFF-C `fl32` / `fl64` / `sowt`); FLAC, OGG and the rest still need
soundfile or ffmpeg.  Scaling
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
AIFF_SUBTYPES = ("PCM_S8", "PCM_16", "PCM_24", "PCM_32", "FLOAT", "DOUBLE")
_PCM, _FLOAT, _EXTENSIBLE = 1, 3, 0xFFFE


def check_format(extension, subtype=None):
    """``soundfile.check_format`` for the formats this build can write."""
    if _sf is not None:
        return _sf.check_format(extension, subtype)
    extension = extension.upper()
    if extension in ("WAV", "WAVE"):
        return subtype is None or subtype in WAV_SUBTYPES
    if extension in ("AIFF", "AIF", "AIFC"):
        return subtype is None or subtype in AIFF_SUBTYPES
    return False


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


# ---------------------------------------------------------------------------
# AIFF / AIFF-C codec (big-endian; the sample rate is an 80-bit IEEE extended float)
# ---------------------------------------------------------------------------
def _extended_from_rate(rate):
    rate = float(rate)
    if rate <= 0:
        return b"\x00" * 10
    exponent = int(np.floor(np.log2(rate)))
    mantissa = int(round(rate / 2.0 ** exponent * (1 << 63)))
    if mantissa >> 64:                                      # rounding carried into the next power of two
        mantissa >>= 1
        exponent += 1
    return struct.pack(">HQ", exponent + 16383, mantissa)


def _rate_from_extended(blob):
    exponent, mantissa = struct.unpack(">HQ", blob)
    if exponent == 0 and mantissa == 0:
        return 0.0
    sign = -1.0 if exponent & 0x8000 else 1.0
    return sign * mantissa / float(1 << 63) * 2.0 ** ((exponent & 0x7FFF) - 16383)


def _pcm_to_float(raw, bits, byteorder):
    if bits == 8:
        return raw.view(np.int8).astype(np.float64) / 128.0
    if bits in (16, 32):
        return raw.view(f"{byteorder}i{bits // 8}").astype(np.float64) / float(1 << (bits - 1))
    if bits == 24:
        b = raw.reshape(-1, 3).astype(np.int32)
        v = (b[:, 0] << 16) | (b[:, 1] << 8) | b[:, 2] if byteorder == ">" else b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - (1 << 24), v)
        return v.astype(np.float64) / float(1 << 23)
    raise RuntimeError(f"Format not recognised: {bits}-bit PCM")


def read_aiff(path):
    """(frames, channels) float64 and the sample rate of an AIFF or AIFF-C file."""
    with open(path, "rb") as fh:
        blob = fh.read()
    if len(blob) < 12 or blob[:4] != b"FORM" or blob[8:12] not in (b"AIFF", b"AIFC"):
        raise RuntimeError("Format not recognised: not an AIFF file")
    compressed = blob[8:12] == b"AIFC"
    comm, sound, pos = None, None, 12
    while pos + 8 <= len(blob):
        tag, size = blob[pos:pos + 4], struct.unpack(">I", blob[pos + 4:pos + 8])[0]
        body = blob[pos + 8:pos + 8 + size]
        if tag == b"COMM":
            comm = body
        elif tag == b"SSND":
            sound = body
        pos += 8 + size + (size & 1)
    if comm is None or sound is None or len(comm) < 18 or len(sound) < 8:
        raise RuntimeError("Format not recognised: missing COMM or SSND chunk")
    channels, frames, bits = struct.unpack(">hIh", comm[:8])
    rate = _rate_from_extended(comm[8:18])
    coding = comm[18:22] if compressed and len(comm) >= 22 else b"NONE"
    offset = struct.unpack(">I", sound[:4])[0]
    data = sound[8 + offset:]
    if channels < 1:
        raise RuntimeError("Format not recognised: bad channel count")
    width = {b"fl32": 4, b"FL32": 4, b"fl64": 8, b"FL64": 8}.get(coding, (bits + 7) // 8)
    frames = min(frames, len(data) // (width * channels))
    raw = np.frombuffer(data, dtype=np.uint8, count=frames * channels * width)
    if coding in (b"fl32", b"FL32"):
        out = raw.view(">f4").astype(np.float64)
    elif coding in (b"fl64", b"FL64"):
        out = raw.view(">f8").astype(np.float64)
    elif coding in (b"NONE", b"twos"):
        out = _pcm_to_float(raw, 8 * width, ">")
    elif coding == b"sowt":                                 # little-endian PCM
        out = _pcm_to_float(raw, 8 * width, "<")
    else:
        raise RuntimeError(f"Format not recognised: AIFF-C compression {coding!r}")
    return out.reshape(frames, channels), int(round(rate))


def write_aiff(path, array, sample_rate, subtype):
    array = np.asarray(array, dtype=np.float64)
    if array.ndim == 1:
        array = array[:, None]
    frames, channels = array.shape
    if subtype in ("FLOAT", "DOUBLE"):
        bits = 32 if subtype == "FLOAT" else 64
        payload = array.astype(">f4" if bits == 32 else ">f8").tobytes()
        coding = (b"fl32", b"32-bit floating point") if bits == 32 else (b"fl64", b"64-bit floating point")
    elif subtype in ("PCM_S8", "PCM_16", "PCM_24", "PCM_32"):
        bits = 8 if subtype == "PCM_S8" else int(subtype[4:])
        top = float((1 << (bits - 1)) - 1)
        q = np.clip(np.rint(array * top), -top - 1.0, top).astype(np.int64)
        if bits == 24:
            u = (q & 0xFFFFFF).astype(np.uint32).reshape(-1)
            b = np.empty((u.size, 3), dtype=np.uint8)
            b[:, 0], b[:, 1], b[:, 2] = (u >> 16) & 0xFF, (u >> 8) & 0xFF, u & 0xFF
            payload = b.tobytes()
        else:
            payload = q.astype({8: "i1", 16: ">i2", 32: ">i4"}[bits]).tobytes()
        coding = None
    else:
        raise TypeError(f"AIFF format does not have {subtype} subtype")
    comm = struct.pack(">hIh", channels, frames, bits) + _extended_from_rate(sample_rate)
    if coding is not None:                                  # AIFF-C: compression type + Pascal string, even length
        name = bytes([len(coding[1])]) + coding[1]
        comm += coding[0] + name + (b"\x00" if len(name) & 1 else b"")
    ssnd = struct.pack(">II", 0, 0) + payload
    chunks = b""
    if coding is not None:
        chunks += b"FVER" + struct.pack(">II", 4, 0xA2805140)
    chunks += b"COMM" + struct.pack(">I", len(comm)) + comm
    chunks += b"SSND" + struct.pack(">I", len(ssnd)) + ssnd + (b"\x00" if len(ssnd) & 1 else b"")
    form = b"AIFC" if coding is not None else b"AIFF"
    with open(path, "wb") as fh:
        fh.write(b"FORM" + struct.pack(">I", 4 + len(chunks)) + form + chunks)


def _read(path):
    if _sf is not None:
        return _sf.read(path, always_2d=True)
    with open(path, "rb") as fh:
        magic = fh.read(12)
    if magic[:4] == b"FORM" and magic[8:12] in (b"AIFF", b"AIFC"):
        return read_aiff(path)
    return read_wav(path)


# ---------------------------------------------------------------------------
# matchering.loader.load / matchering.saver.save
# ---------------------------------------------------------------------------
def load(file: str, file_type: str, temp_folder: str):
    """loader.py:30-47: returns ``(sound (n, channels) float64, sample_rate)``; raises
    ``ModuleError(4001 | 4101)`` when the file cannot be decoded (after trying ffmpeg)."""
    file_type = file_type.upper()
    sound, sample_rate = None, None
    debug(f"reading {file_type} from '{file}'")
    try:
        sound, sample_rate = _read(file)
    except (RuntimeError, OSError) as e:
        debug(e)
        if "unknown format" in str(e) or "Format not recognised" in str(e):
            sound, sample_rate = _load_with_ffmpeg(file, file_type, temp_folder)
    if sound is None or sample_rate is None:
        raise ModuleError(Code.ERROR_TARGET_LOADING if file_type == "TARGET" else Code.ERROR_REFERENCE_LOADING)
    debug(f"{file_type}: {sound.shape[0]} frames, {sound.shape[1]} channel(s) at {sample_rate} Hz")
    return sound, sample_rate


def _load_with_ffmpeg(file, file_type, temp_folder):
    """loader.py:50-74: decode through an ``ffmpeg`` subprocess into a temporary WAV."""
    sound, sample_rate = None, None
    debug(f"no built-in decoder for '{file}': handing it to ffmpeg")
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
            debug("ffmpeg is not on PATH, so formats beyond the built-in codecs cannot be read")
        except subprocess.CalledProcessError:
            debug(f"ffmpeg could not decode '{file}' either")
    return sound, sample_rate


def save(file: str, result: np.ndarray, sample_rate: int, subtype: str, name: str = "result") -> None:
    """saver.py:27-33."""
    debug(f"writing the {name} as {subtype} at {sample_rate} Hz to '{file}'")
    if _sf is not None:
        _sf.write(file, result, sample_rate, subtype)
    elif os.path.splitext(file)[1][1:].upper() in ("AIFF", "AIF", "AIFC"):
        write_aiff(file, result, sample_rate, subtype)
    else:
        write_wav(file, result, sample_rate, subtype)
    debug(f"wrote '{file}'")
