"""Every knob of the reference's Config is accepted (cf. matchering's examples/edited_config.py); the
ones that reach the GPU are passed to libmgx in `mgx_config`.  Values the kernels do not implement
(fft_size above 16384, limiter filter orders above 1, lowess_it above 0) raise instead of being
approximated."""
import matchering_amd as mg

config = mg.Config(
    max_length=30 * 60,              # accept half-hour tracks (default 15 minutes)
    internal_sample_rate=96000,      # work and export at 96 kHz (default 44.1 kHz)
    fft_size=16384,                  # a 16 k-tap matching FIR (default 4096)
    threshold=0.7079,                # limit to -3 dBFS (default -0.01 dBFS)
    preview_size=15,                 # seconds
    allow_equality=True,             # do not refuse target == reference
    limiter=mg.LimiterConfig(attack=2.0, hold=2.0, release=1500.0),
)

mg.process(
    target="my_song.wav",
    reference="some_popular_song.wav",
    results=[mg.pcm24("my_song_master_24bit_96k.wav")],
    config=config,
)
