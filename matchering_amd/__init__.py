"""matchering_amd -- MI355X-native mastering core behind the matchering API.

Public names mirror matchering/__init__.py:31-36 (``log``, ``Result``, ``pcm16``,
``pcm24``, ``Config``, ``process``, ``load``, ``check``); the DSP of
``matchering.stages.main`` runs as hand-written HIP kernels (libmgx.so).
"""

__version__ = "0.1.0"
__reference_version__ = "2.0.6"

from .log import set_handlers as log  # noqa: F401
from .results import Result, pcm16, pcm24  # noqa: F401
from .config import Config, LimiterConfig  # noqa: F401
from .core import process  # noqa: F401
from .audio_io import load  # noqa: F401
from .checker import check  # noqa: F401
from .batch import master_many, process_batch  # noqa: F401  (no counterpart in the reference)
