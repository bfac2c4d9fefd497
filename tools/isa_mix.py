"""Static instruction mix of one kernel in a device assembly listing.

    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S -o mgx.s matchering_amd/csrc/mgx.hip
    python tools/isa_mix.py mgx.s _ZN3mgx6k_convILi13ELb0ELi1EEEvNS_9Conv2ArgsE

Prints instruction counts by class for the whole kernel and per basic block (largest first), so a
change to a phase can be judged by the VALU / LDS / VMEM instructions it adds or removes before a
GPU run is spent on it.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    blocks = collections.OrderedDict()
    cur = "entry"
    blocks[cur] = collections.Counter()
    detail = collections.Counter()
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = collections.Counter()
            continue
        s = l.strip()
        if not s or s.startswith((";", ".")):
            continue
        op = s.split()[0]
        blocks[cur][classify(op)] += 1
        detail[op] += 1
    total = collections.Counter()
    for c in blocks.values():
        total.update(c)
    print("total", dict(total))
    for b, c in sorted(blocks.items(), key=lambda kv: -sum(kv[1].values()))[:12]:
        print(f"{b:14s} {sum(c.values()):6d}  " + "  ".join(f"{k}={v}" for k, v in sorted(c.items())))
    print("top ops:", ", ".join(f"{k}={v}" for k, v in detail.most_common(28)))


if __name__ == "__main__":
    main()
