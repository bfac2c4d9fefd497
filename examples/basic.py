"""matchering's examples/basic.py on the MI355X path: only the import changes."""
import sys

import matchering_amd as mg

# every log level to stdout; drop the line for a silent run
mg.log(print)

target, reference = (sys.argv[1:3] + ["my_song.wav", "some_popular_song.wav"])[:2]
mg.process(
    target=target,
    reference=reference,
    results=[
        mg.pcm16("my_song_master_16bit.wav"),
        mg.pcm24("my_song_master_24bit.wav"),
    ],
)
