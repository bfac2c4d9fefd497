"""Status codes and their English texts in one table (numbering and wording follow
matchering/log/codes.py:24-58 and log/explanations.py:36-71 so that applications
which parse the codes keep working)."""

from enum import IntEnum

_T, _R = "TARGET", "REFERENCE"
_TABLE = (
    # name, value, text
    ("INFO_UPLOADING", 2001, "Uploading files"),
    ("INFO_WAITING", 2002, "Queued for processing"),
    ("INFO_LOADING", 2003, "Loading and analysis"),
    ("INFO_MATCHING_LEVELS", 2004, "Matching levels"),
    ("INFO_MATCHING_FREQS", 2005, "Matching frequencies"),
    ("INFO_CORRECTING_LEVELS", 2006, "Correcting levels"),
    ("INFO_FINALIZING", 2007, "Final processing and saving"),
    ("INFO_EXPORTING", 2008, "Exporting various audio formats"),
    ("INFO_MAKING_PREVIEWS", 2009, "Making previews"),
    ("INFO_COMPLETED", 2010, "The task is completed"),
    ("INFO_TARGET_IS_MONO", 2101, f"The {_T} audio is mono. Converting it to stereo..."),
    ("INFO_REFERENCE_IS_MONO", 2201, f"The {_R} audio is mono. Converting it to stereo..."),
    ("INFO_REFERENCE_IS_RESAMPLED", 2202, f"The {_R} audio was resampled"),
    ("INFO_REFERENCE_IS_LOSSY", 2203, f"Presumably the {_R} audio format is lossy"),
    ("WARNING_TARGET_IS_CLIPPING", 3001,
     f"Audio clipping is detected in the {_T} file. It is highly recommended to use the non-clipping version"),
    ("WARNING_TARGET_LIMITER_IS_APPLIED", 3002,
     f"The applied limiter is detected in the {_T} file. "
     "It is highly recommended to use the version without a limiter"),
    ("WARNING_TARGET_IS_RESAMPLED", 3003,
     f"The {_T} audio sample rate and internal sample rate were different. The {_T} audio was resampled"),
    ("WARNING_TARGET_IS_LOSSY", 3004,
     f"Presumably the {_T} audio format is lossy. "
     "It is highly recommended to use lossless audio formats (WAV, FLAC, AIFF)"),
    ("ERROR_TARGET_LOADING", 4001, f"Audio stream error in the {_T} file"),
    ("ERROR_TARGET_LENGTH_IS_EXCEEDED", 4002, f"Track length is exceeded in the {_T} file"),
    ("ERROR_TARGET_LENGTH_IS_TOO_SMALL", 4003, f"The track length is too small in the {_T} file"),
    ("ERROR_TARGET_NUM_OF_CHANNELS_IS_EXCEEDED", 4004, f"The number of channels exceeded in the {_T} file"),
    ("ERROR_TARGET_EQUALS_REFERENCE", 4005,
     f"The {_T} and {_R} files are the same. They must be different so that Matchering makes sense"),
    ("ERROR_REFERENCE_LOADING", 4101, f"Audio stream error in the {_R} file"),
    ("ERROR_REFERENCE_LENGTH_LENGTH_IS_EXCEEDED", 4102, f"Track length is exceeded in the {_R} file"),
    ("ERROR_REFERENCE_LENGTH_LENGTH_TOO_SMALL", 4103, f"The track length is too small in the {_R} file"),
    ("ERROR_REFERENCE_NUM_OF_CHANNELS_IS_EXCEEDED", 4104, f"The number of channels exceeded in the {_R} file"),
    ("ERROR_UNKNOWN", 4201, "Unknown error"),
    ("ERROR_VALIDATION", 4202, "Validation failed! Please let the developers know about this error!"),
)

Code = IntEnum("Code", [(name, value) for name, value, _ in _TABLE])
_TEXT = {Code[name]: text for name, _, text in _TABLE}


def explain(code, with_code=False):
    """Human-readable text of a code; ``with_code`` prefixes ``"<number>: "``."""
    text = _TEXT[Code(code)]
    return f"{int(code)}: {text}" if with_code else text
