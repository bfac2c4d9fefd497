"""Callback logging with numeric status codes -- the error/progress convention of
the boundary (matchering/log/*: ``mg.log(...)`` installs handlers, stages emit
``info(Code.X)``, failures raise ``ModuleError(Code.Y)``)."""

from .catalog import Code, explain
from .sinks import debug, debug_line, info, set_handlers, warning


class ModuleError(Exception):
    """Raised for every reference-defined failure; message is ``"<code>: <text>"``
    (log/exceptions.py:25-27)."""

    def __init__(self, code: Code):
        self.code = code
        super().__init__(explain(code, with_code=True))


__all__ = ["Code", "ModuleError", "explain", "set_handlers", "info", "warning", "debug", "debug_line"]
