"""Small host helpers (matchering/utils.py:46-59 equivalents used on the boundary)."""

import math
from datetime import timedelta


def to_db(value: float) -> str:
    return f"{20 * math.log10(value):.4f} dB"


def ms_to_samples(value: float, sample_rate: int) -> int:
    return int(sample_rate * value * 1e-3)


def make_odd(value: int) -> int:
    return value | 1


def time_str(length, sample_rate) -> str:
    return str(timedelta(seconds=length // sample_rate))
