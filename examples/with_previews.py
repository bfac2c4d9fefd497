"""Ask for two short previews next to the results (cf. matchering's examples/with_preview.py): the loudest
`preview_size` seconds of the target and the same window of the master, faded in and out, for a
quick before/after comparison."""
import matchering_amd as mg

mg.log(warning_handler=print)

mg.process(
    target="my_song.wav",
    reference="some_popular_song.wav",
    results=[mg.pcm16("my_song_master_16bit.wav")],
    preview_target=mg.pcm16("preview_my_song.wav"),
    preview_result=mg.pcm16("preview_my_song_master.wav"),
)
