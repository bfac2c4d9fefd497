"""Where a kernel spills: scratch stores / reloads, global loads / stores and barriers per basic
block of a device assembly listing (see tools/isa_mix.py for how to produce one).

    python tools/spill_map.py mgx.s <mangled kernel name>
"""
import itertools
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
name = sys.argv[2]
st = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
seq = []
for l in lines[st + 1:]:
    if l.startswith(".Lfunc_end"):
        break
    s = l.strip()
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        seq.append(("label", m.group(1)))
    elif s.startswith("s_barrier"):
        seq.append(("barrier",))
    elif s.startswith("scratch_store"):
        seq.append(("spill",))
    elif s.startswith("scratch_load"):
        seq.append(("reload",))
    elif s.startswith(("buffer_load", "global_load")):
        seq.append(("gld",))
    elif s.startswith(("buffer_store", "global_store")):
        seq.append(("gst",))
out = []
for k, g in itertools.groupby(seq):
    n = len(list(g))
    out.append(f"\n{k[1]}:" if k[0] == "label" else f"{k[0]}x{n}")
print(" ".join(out))
