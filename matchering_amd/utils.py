"""Small host helpers (matchering/utils.py:46-59 equivalents used on the boundary)."""

import math
import os
import random
import string
from datetime import timedelta


def to_db(value: float) -> str:
    return f"{20 * math.log10(value):.4f} dB"


def ms_to_samples(value: float, sample_rate: int) -> int:
    return int(sample_rate * value * 1e-3)


def make_odd(value: int) -> int:
    return value | 1


def time_str(length, sample_rate) -> str:
    return str(timedelta(seconds=length // sample_rate))


def get_temp_folder(results: list) -> str:
    """utils.py:28-30: the directory of the first requested result."""
    return os.path.dirname(os.path.abspath(results[0].file))


def random_file(prefix: str = "", extension: str = "wav") -> str:
    """utils.py:33-39: ``<prefix>-<16 random chars>.<extension>``."""
    stem = "".join(random.choices(string.ascii_lowercase + string.digits, k=16))
    return f"{prefix + '-' if prefix else ''}{stem}.{extension}"
