"""Many pairs at once -- something the reference leaves to matchering-cli.  One process per GPU, e.g.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/batch_of_pairs.py

(plain `python examples/batch_of_pairs.py` uses one GPU).  Pair i is mastered by rank i mod world size;
inside a rank two device handles keep two pairs in flight while loader and writer threads decode the
next files and encode finished ones."""
import matchering_amd as mg

jobs = [
    {"target": f"album/track_{i:02d}.wav", "reference": "some_popular_song.wav",
     "results": [mg.pcm24(f"album/master_{i:02d}.wav")]}
    for i in range(1, 13)
]

done = mg.process_batch(jobs, mg.Config(), lanes=2, io_threads=4)
print(f"this rank mastered pairs {done}")
