"""What to write and how: ``Result`` and its two shortcuts (matchering/results.py:25-46).

A ``Result`` names a file, the sample format inside it and which rendering of the master it
receives: ``use_limiter=True`` the limited one; otherwise the matched track, peak-normalised to the
threshold when ``normalize`` is set and left as it is (possibly above 0 dBFS) when not.
"""

import os

from .audio_io import check_format


def _container_of(path):
    return os.path.splitext(path)[1].lstrip(".").upper()


class Result:
    __slots__ = ("file", "subtype", "use_limiter", "normalize")

    def __init__(self, file: str, subtype: str, use_limiter: bool = True, normalize: bool = True):
        container = _container_of(file)
        # same two TypeErrors, in the same order, as the reference raises through soundfile.check_format
        if not check_format(container):
            raise TypeError(f"{container} format is not supported")
        if not check_format(container, subtype):
            raise TypeError(f"{container} format does not have {subtype} subtype")
        self.file, self.subtype = file, subtype
        self.use_limiter, self.normalize = use_limiter, normalize

    def __repr__(self):
        return (f"Result({self.file!r}, {self.subtype!r}, use_limiter={self.use_limiter}, "
                f"normalize={self.normalize})")


def pcm16(file: str) -> Result:
    """16-bit integer samples, limited master."""
    return Result(file, subtype="PCM_16")


def pcm24(file: str) -> Result:
    """24-bit integer samples, limited master."""
    return Result(file, subtype="PCM_24")
