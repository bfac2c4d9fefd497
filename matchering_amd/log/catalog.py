"""Status codes and what they mean.

The NAMES and NUMBERS are the reference's (matchering/log/codes.py:24-58): applications built on
matchering switch on them, so they are part of the API this package mirrors.  The English texts are
this package's own wording; they are composed from a handful of templates because most codes exist
once for the TARGET and once for the REFERENCE (4001/4101, 4002/4102, ...).
"""

from enum import IntEnum

# progress of one process() call, in the order it is reported
_PROGRESS = {
    2001: ("INFO_UPLOADING", "Receiving the files"),
    2002: ("INFO_WAITING", "Waiting in the queue"),
    2003: ("INFO_LOADING", "Reading and checking the audio"),
    2004: ("INFO_MATCHING_LEVELS", "Matching the loudness"),
    2005: ("INFO_MATCHING_FREQS", "Matching the spectrum"),
    2006: ("INFO_CORRECTING_LEVELS", "Re-adjusting the loudness"),
    2007: ("INFO_FINALIZING", "Limiting and finishing"),
    2008: ("INFO_EXPORTING", "Writing the result files"),
    2009: ("INFO_MAKING_PREVIEWS", "Cutting the previews"),
    2010: ("INFO_COMPLETED", "Done"),
}

# per-track conditions: offset within the track's block -> (name pattern, severity prefix, template)
_TRACKS = {"TARGET": 0, "REFERENCE": 100}
_FAILURES = {
    1: ("{t}_LOADING", "the {t} file could not be decoded"),
    2: ("{t}_LENGTH{x}_IS_EXCEEDED", "the {t} track is longer than max_length allows"),
    3: ("{t}_LENGTH{x}_TOO_SMALL", "the {t} track is shorter than one analysis window"),
    4: ("{t}_NUM_OF_CHANNELS_IS_EXCEEDED", "the {t} file has more than two channels"),
}


def _build():
    rows = [(number, name, text) for number, (name, text) in _PROGRESS.items()]
    rows += [
        (2101, "INFO_TARGET_IS_MONO", "the TARGET is mono: both channels get the same signal"),
        (2201, "INFO_REFERENCE_IS_MONO", "the REFERENCE is mono: both channels get the same signal"),
        (2202, "INFO_REFERENCE_IS_RESAMPLED", "the REFERENCE was converted to the internal sample rate"),
        (2203, "INFO_REFERENCE_IS_LOSSY", "the REFERENCE seems to come from a lossy codec"),
        (3001, "WARNING_TARGET_IS_CLIPPING",
         "the TARGET clips (runs of samples at full scale): master from an unclipped mix if you can"),
        (3002, "WARNING_TARGET_LIMITER_IS_APPLIED",
         "the TARGET looks limited already: master from the version without a limiter if you can"),
        (3003, "WARNING_TARGET_IS_RESAMPLED",
         "the TARGET was not at the internal sample rate and has been converted"),
        (3004, "WARNING_TARGET_IS_LOSSY",
         "the TARGET seems to come from a lossy codec: prefer WAV, FLAC or AIFF sources"),
    ]
    for track, base in _TRACKS.items():
        for offset, (pattern, template) in _FAILURES.items():
            # the reference spells the REFERENCE length errors with a doubled LENGTH and, for the second
            # one, without IS: ERROR_REFERENCE_LENGTH_LENGTH_IS_EXCEEDED / ..._LENGTH_LENGTH_TOO_SMALL
            extra = "_LENGTH" if track == "REFERENCE" else ""
            name = "ERROR_" + pattern.format(t=track, x=extra)
            if track == "TARGET" and offset == 3:
                name = "ERROR_TARGET_LENGTH_IS_TOO_SMALL"
            rows.append((4000 + base + offset, name, template.format(t=track)))
    rows += [
        (4005, "ERROR_TARGET_EQUALS_REFERENCE", "TARGET and REFERENCE are the same audio: there is nothing to match"),
        (4201, "ERROR_UNKNOWN", "an error without a code of its own"),
        (4202, "ERROR_VALIDATION", "the loaded audio failed the final consistency check (please report this)"),
    ]
    return sorted(rows)


_ROWS = _build()
Code = IntEnum("Code", [(name, number) for number, name, _ in _ROWS])
_TEXT = {Code(number): text[0].upper() + text[1:] for number, _, text in _ROWS}


def explain(code, with_code=False):
    """Text of a code; ``with_code`` puts ``"<number>: "`` in front (log/explanations.py:28-29)."""
    text = _TEXT[Code(code)]
    return f"{int(code)}: {text}" if with_code else text
