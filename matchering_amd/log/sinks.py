"""Where log lines go.  ``set_handlers`` (exported as ``mg.log``) has the signature
of matchering/log/handlers.py:54-67: one default callable plus optional per-level
overrides, ``show_codes`` to prefix the numeric code."""

from .catalog import explain


def _drop(*_args, **_kwargs):
    return None


_state = {"warning": _drop, "info": _drop, "debug": _drop, "show_codes": False}


def set_handlers(default_handler=None, warning_handler=None, info_handler=None, debug_handler=None,
                 show_codes=False):
    fallback = default_handler or _drop
    _state["warning"] = warning_handler or fallback
    _state["info"] = info_handler or fallback
    _state["debug"] = debug_handler or fallback
    _state["show_codes"] = bool(show_codes)


def warning(code):
    _state["warning"](explain(code, _state["show_codes"]))


def info(code):
    _state["info"](explain(code, _state["show_codes"]))


def debug(*args, **kwargs):
    _state["debug"](*args, **kwargs)


def debug_line():
    debug("-" * 40)
