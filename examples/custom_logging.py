"""Route the three log levels wherever you like (cf. matchering's examples/advanced_text_output.py):
INFO carries the progress codes 2001-2010, WARNING the codes 3001-3004 about the target (clipping,
limiting, lossy source, resampling), DEBUG the per-stage scalars read back from the GPU."""
import time

import matchering_amd as mg

START = time.time()


def stamped(level):
    return lambda text: print(f"[{time.time() - START:8.3f} s] {level:7s} {text}")


mg.log(info_handler=stamped("INFO"), warning_handler=stamped("WARNING"), debug_handler=stamped("DEBUG"))

mg.process(
    target="my_song.wav",
    reference="some_popular_song.wav",
    results=[mg.pcm16("my_song_master_16bit.wav"), mg.pcm24("my_song_master_24bit.wav")],
)
