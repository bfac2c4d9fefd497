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

Integer PCM can also pass through undecoded (``load(..., pcm=True)``, integer arrays given to ``save``):
the samples then cross PCIe as the file holds them and are converted on the GPU (``mgx_pcm_decode`` /
``mgx_pcm_encode``, same scaling).  Such an array is int16 (n, channels), int32 (n, channels), or uint8
(n, channels * 3) for packed little-endian 24-bit samples; ``pcm_to_float`` turns any of them into the
floats the default path returns.
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
def _wav_layout(path):
    """Header walk of a RIFF/WAVE file: (format code, channels, rate, block size, bits, data offset, data bytes)."""
    with open(path, "rb") as fh:
        head = fh.read(12)
        if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise RuntimeError("Format not recognised: not a RIFF/WAVE file")
        size_of_file = os.fstat(fh.fileno()).st_size
        fmt, pos = None, 12
        while pos + 8 <= size_of_file:
            fh.seek(pos)
            tag_size = fh.read(8)
            if len(tag_size) < 8:
                break
            tag, size = tag_size[:4], struct.unpack("<I", tag_size[4:])[0]
            if tag == b"fmt ":
                fmt = fh.read(size)
            elif tag == b"data":
                if fmt is None:                 # a data chunk ahead of fmt: keep looking for fmt, come back
                    data_at, data_size = pos + 8, size
                    pos += 8 + size + (size & 1)
                    while pos + 8 <= size_of_file and fmt is None:
                        fh.seek(pos)
                        t2 = fh.read(8)
                        tag2, size2 = t2[:4], struct.unpack("<I", t2[4:])[0]
                        if tag2 == b"fmt ":
                            fmt = fh.read(size2)
                        pos += 8 + size2 + (size2 & 1)
                    pos, size = data_at - 8, data_size
                if fmt is None or len(fmt) < 16:
                    break
                code, channels, rate, _, block, bits = struct.unpack("<HHIIHH", fmt[:16])
                if code == _EXTENSIBLE and len(fmt) >= 26:
                    code = struct.unpack("<H", fmt[24:26])[0]       # first two bytes of the sub-format GUID
                if channels < 1 or block < 1:
                    raise RuntimeError("Format not recognised: bad channel count")
                if code in (_PCM, _FLOAT) and block != channels * ((bits + 7) // 8):
                    raise RuntimeError(f"Format not recognised: block size {block} for {channels} channel(s) of {bits} bits")
                return code, channels, int(rate), block, bits, pos + 8, min(size, size_of_file - pos - 8)
            pos += 8 + size + (size & 1)
    raise RuntimeError("Format not recognised: missing fmt or data chunk")


def unpack24(array):
    """Packed little-endian 24-bit samples, uint8 (n, channels * 3) -> int32 (n, channels) with the sample in
    the three high bytes (the value at scale 2**31): one strided copy."""
    frames = array.shape[0]
    wide = np.zeros((array.size // 3, 4), dtype=np.uint8)
    wide[:, 1:] = np.asarray(array).reshape(-1, 3)
    return wide.view("<i4").reshape(frames, -1)


def pcm_channels(array):
    """Channel count of (n, channels) audio, or of packed 24-bit PCM (n, channels * 3) uint8."""
    return array.shape[1] // 3 if array.dtype == np.uint8 else array.shape[1]


def pcm_to_float(array, dtype=np.float32):
    """Integer PCM as ``load(..., pcm=True)`` returns it -> floats in [-1, 1): ``v / 2**(bits-1)``, exact
    in float32 up to 24 bits.  Float arrays pass through."""
    dtype = np.dtype(dtype).type
    if array.dtype == np.int16:
        out = array.astype(dtype)
        out *= dtype(1.0 / 32768.0)
        return out
    if array.dtype == np.int32:
        out = array.astype(np.float64)
        out *= 1.0 / 2147483648.0
        return out.astype(dtype, copy=False)
    if array.dtype == np.uint8:                            # packed little-endian 24-bit
        out = (unpack24(array) >> 8).astype(dtype)
        out *= dtype(1.0 / 8388608.0)
        return out
    return array


def read_wav(path, pcm=False):
    """(frames, channels) samples and the sample rate.  Floats in [-1, 1): float32 for integer files up to
    24 bits (exact), float64 for 32-bit integer and for DOUBLE files, float32 for FLOAT files.  With
    ``pcm=True`` files of 16, 24 or 32-bit integers come back undecoded, as a read-only view of the file
    (module docstring; 24-bit samples packed, uint8 (frames, channels * 3))."""
    code, channels, rate, block, bits, offset, nbytes = _wav_layout(path)
    frames = nbytes // block
    if pcm and frames > 0 and ((code == _PCM and bits in (16, 24, 32)) or (code == _FLOAT and bits == 32)):
        # undecoded samples are only read (checks, staging for the upload): map the file instead of copying it
        raw = np.memmap(path, dtype=np.uint8, mode="r", offset=offset, shape=(frames * block,))
    else:
        raw = np.fromfile(path, dtype=np.uint8, count=frames * block, offset=offset)
    if code == _FLOAT and bits in (32, 64):
        out = raw.view("<f4" if bits == 32 else "<f8")
    elif code == _PCM and bits == 8:
        out = (raw.astype(np.float32) - 128.0) / 128.0
    elif code == _PCM and bits == 16:
        out = raw.view("<i2").reshape(frames, channels)
        if not pcm:
            out = pcm_to_float(out)
    elif code == _PCM and bits == 32:
        out = raw.view("<i4").reshape(frames, channels)
        if not pcm:
            out = pcm_to_float(out, np.float64)
    elif code == _PCM and bits == 24:
        out = raw.reshape(frames, channels * 3)                    # packed: three bytes per sample
        if not pcm:                                                # 24 bits are exact in float32
            out = (unpack24(out) >> 8).astype(np.float32)
            out *= np.float32(1.0 / 8388608.0)
    else:
        raise RuntimeError(f"Format not recognised: WAVE format tag {code} with {bits} bits")
    return out.reshape(frames, -1) if out.dtype == np.uint8 else out.reshape(frames, channels), rate


def _quantise(array, bits):
    """rint(x * (2**(bits-1) - 1)) clipped to the integer range, in float64 blocks (bounded temporaries)."""
    top = float((1 << (bits - 1)) - 1)
    flat = np.ascontiguousarray(array).reshape(-1)
    out = np.empty(flat.shape[0], dtype=np.int16 if bits <= 16 else np.int32)
    step = 1 << 20
    for i in range(0, flat.shape[0], step):
        q = flat[i:i + step].astype(np.float64)
        q *= top
        np.rint(q, out=q)
        np.clip(q, -top - 1.0, top, out=q)
        out[i:i + step] = q
    return out


def _pack24(q, big_endian=False):
    """int32 samples -> packed 3-byte samples (uint8, three per sample)."""
    b = np.empty((q.shape[0], 3), dtype=np.uint8)
    lo, mid, hi = (q & 0xFF), ((q >> 8) & 0xFF), ((q >> 16) & 0xFF)
    if big_endian:
        b[:, 0], b[:, 1], b[:, 2] = hi, mid, lo
    else:
        b[:, 0], b[:, 1], b[:, 2] = lo, mid, hi
    return b


def _pcm_matches(array, bits):
    return (bits == 16 and array.dtype == np.int16) or (bits == 32 and array.dtype == np.int32) or \
           (bits == 24 and array.dtype == np.uint8)


def write_wav(path, array, sample_rate, subtype):
    """``array``: floats (n, channels), or integer PCM already quantised for this ``subtype`` (int16 for
    PCM_16, int32 for PCM_32, uint8 (n, channels * 3) for PCM_24: what the device encoder hands back)."""
    array = np.asarray(array)
    if array.ndim == 1:
        array = array[:, None]
    frames = array.shape[0]
    if subtype in ("FLOAT", "DOUBLE"):
        code, bits = _FLOAT, 32 if subtype == "FLOAT" else 64
        channels = array.shape[1]
        payload = np.ascontiguousarray(array, dtype="<f4" if bits == 32 else "<f8")
    elif subtype == "PCM_U8":
        code, bits, channels = _PCM, 8, array.shape[1]
        payload = np.clip(np.rint(array.astype(np.float64) * 127.0) + 128.0, 0, 255).astype(np.uint8)
    elif subtype in ("PCM_16", "PCM_24", "PCM_32"):
        code, bits = _PCM, int(subtype[4:])
        if _pcm_matches(array, bits):
            channels = array.shape[1] // 3 if bits == 24 else array.shape[1]
            payload = np.ascontiguousarray(array)
        elif array.dtype.kind in "iu":
            raise TypeError(f"integer samples of type {array.dtype} cannot be written as {subtype}")
        else:
            channels = array.shape[1]
            q = _quantise(array, bits)
            payload = _pack24(q) if bits == 24 else q.astype("<i2" if bits == 16 else "<i4", copy=False)
    else:
        raise TypeError(f"WAV format does not have {subtype} subtype")
    block = channels * bits // 8
    size = frames * block
    fmt = struct.pack("<HHIIHH", code, channels, int(sample_rate), int(sample_rate) * block, block, bits)
    head = b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if code == _FLOAT:
        head += b"fact" + struct.pack("<II", 4, frames)
    head += b"data" + struct.pack("<I", size)
    pad = size & 1
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", 4 + len(head) + size + pad) + b"WAVE" + head)
        fh.write(memoryview(payload).cast("B"))
        if pad:
            fh.write(b"\x00")


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
    array = pcm_to_float(np.asarray(array), np.float64)          # (integer PCM is little-endian: decode, re-encode)
    if array.ndim == 1:
        array = array[:, None]
    frames, channels = array.shape
    if subtype in ("FLOAT", "DOUBLE"):
        bits = 32 if subtype == "FLOAT" else 64
        payload = array.astype(">f4" if bits == 32 else ">f8").tobytes()
        coding = (b"fl32", b"32-bit floating point") if bits == 32 else (b"fl64", b"64-bit floating point")
    elif subtype in ("PCM_S8", "PCM_16", "PCM_24", "PCM_32"):
        bits = 8 if subtype == "PCM_S8" else int(subtype[4:])
        q = _quantise(array, bits)
        if bits == 24:
            payload = _pack24(q, big_endian=True).tobytes()
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


def _native_wave(path):
    """True for the WAVE layouts ``read_wav`` decodes itself (and can hand on undecoded): integer PCM of 8,
    16, 24 or 32 bits, FLOAT of 32 or 64.  Everything else libsndfile reads -- A-law, mu-law, ADPCM, GSM,
    odd containers -- stays with libsndfile when it is installed (loader.py:35)."""
    try:
        code, _, _, _, bits, _, _ = _wav_layout(path)
    except (RuntimeError, OSError, struct.error, ValueError):
        return False
    return (code == _PCM and bits in (8, 16, 24, 32)) or (code == _FLOAT and bits in (32, 64))


def _read(path, pcm=False):
    with open(path, "rb") as fh:
        magic = fh.read(12)
    wave = magic[:4] == b"RIFF" and magic[8:12] == b"WAVE"
    if wave and (_sf is None or (pcm and _native_wave(path))):
        try:
            return read_wav(path, pcm)
        except (ValueError, struct.error) as exc:          # a header that lies about its sizes
            if _sf is None:
                raise RuntimeError(f"Format not recognised: {exc}") from exc
    if _sf is not None:
        return _sf.read(path, always_2d=True)
    if magic[:4] == b"FORM" and magic[8:12] in (b"AIFF", b"AIFC"):
        return read_aiff(path)
    return read_wav(path, pcm)


# ---------------------------------------------------------------------------
# matchering.loader.load / matchering.saver.save
# ---------------------------------------------------------------------------
def load(file: str, file_type: str, temp_folder: str, pcm: bool = False):
    """loader.py:30-47: returns ``(sound (n, channels) floats, sample_rate)``; raises
    ``ModuleError(4001 | 4101)`` when the file cannot be decoded (after trying ffmpeg).  ``pcm=True``
    (what ``process`` passes) leaves WAVE files of 16, 24 or 32-bit integers undecoded (module docstring)."""
    file_type = file_type.upper()
    sound, sample_rate = None, None
    debug(f"reading {file_type} from '{file}'")
    try:
        sound, sample_rate = _read(file, pcm)
    except (RuntimeError, OSError) as e:
        debug(e)
        if "unknown format" in str(e) or "Format not recognised" in str(e):
            sound, sample_rate = _load_with_ffmpeg(file, file_type, temp_folder)
    if sound is None or sample_rate is None:
        raise ModuleError(Code.ERROR_TARGET_LOADING if file_type == "TARGET" else Code.ERROR_REFERENCE_LOADING)
    debug(f"{file_type}: {sound.shape[0]} frames, {pcm_channels(sound)} channel(s) at {sample_rate} Hz")
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
    extension = os.path.splitext(file)[1][1:].upper()
    if np.asarray(result).dtype.kind in "iu" and extension in ("WAV", "WAVE"):
        write_wav(file, result, sample_rate, subtype)              # quantised on the GPU already
    elif _sf is not None:
        _sf.write(file, pcm_to_float(np.asarray(result), np.float64), sample_rate, subtype)
    elif extension in ("AIFF", "AIF", "AIFC"):
        write_aiff(file, result, sample_rate, subtype)
    else:
        write_wav(file, result, sample_rate, subtype)
    debug(f"wrote '{file}'")
