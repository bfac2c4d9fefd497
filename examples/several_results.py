"""One mastering run, several renderings of it (cf. matchering's examples/advanced_results.py).

`Result(file, subtype, use_limiter, normalize)` picks which of the three outputs of stages.main a
file gets: the limited master, the matched-but-unlimited track normalised to the threshold, or the
same without normalisation (it may exceed 0 dBFS, so store it as float).  WAV and AIFF are written by
the built-in codecs; FLAC, OGG, ... need the optional `soundfile` package.
"""
import matchering_amd as mg

mg.log(info_handler=print, warning_handler=print)

mg.process(
    target="my_song.wav",
    reference="some_popular_song.wav",
    results=[
        mg.pcm16("my_song_master_16bit.wav"),                                    # match + limiter
        mg.Result("my_song_matched_24bit.wav", subtype="PCM_24", use_limiter=False),   # no limiter, peak at the threshold
        mg.Result("my_song_matched_float.aiff", subtype="FLOAT", use_limiter=False, normalize=False),   # raw match
    ],
)
