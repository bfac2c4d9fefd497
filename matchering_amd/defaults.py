"""Import-path compatibility: ``from matchering.defaults import Config, LimiterConfig``."""
from .config import Config, LimiterConfig  # noqa: F401
