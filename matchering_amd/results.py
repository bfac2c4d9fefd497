"""``Result`` / ``pcm16`` / ``pcm24`` (matchering/results.py:25-46): what to write and how."""

import os

from .audio_io import check_format


class Result:
    def __init__(self, file: str, subtype: str, use_limiter: bool = True, normalize: bool = True):
        extension = os.path.splitext(file)[1][1:].upper()
        if not check_format(extension):
            raise TypeError(f"{extension} format is not supported")
        if not check_format(extension, subtype):
            raise TypeError(f"{extension} format does not have {subtype} subtype")
        self.file = file
        self.subtype = subtype
        self.use_limiter = use_limiter
        self.normalize = normalize


def pcm16(file: str) -> Result:
    return Result(file, "PCM_16")


def pcm24(file: str) -> Result:
    return Result(file, "PCM_24")
