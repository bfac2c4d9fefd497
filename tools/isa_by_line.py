"""Instructions of one kernel attributed to source lines (a -gline-tables-only device listing).

    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -gline-tables-only -S -o mgx_g.s matchering_amd/csrc/mgx.hip
    python tools/isa_by_line.py mgx_g.s _ZN3mgx7k_limitILi256ELi4EEEvNS_11LimiterArgsE [limiter_kernel.h]

Counts the instructions that follow each `.loc file line` inside the kernel and prints them per file and line
(largest first), then per function span when a source file is named: where a kernel's code bytes come from.
"""
import collections
import re
import sys


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    want_file = sys.argv[3] if len(sys.argv) > 3 else None
    files = {}
    counts = collections.Counter()
    inside = False
    cur = None
    for line in open(path):
        s = line.strip()
        m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s) or re.match(r'\.file\s+(\d+)\s+"([^"]+)"', s)
        if m:
            files[int(m.group(1))] = m.group(2).split("/")[-1]
            continue
        if s.startswith(kernel + ":"):
            inside = True
            continue
        if inside and s.startswith(".Lfunc_end"):
            break
        if not inside:
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith((".", ";")) or s.endswith(":"):
            continue
        counts[cur] += 1
    total = sum(counts.values())
    print("total instructions", total)
    per_file = collections.Counter()
    for (f, _l), c in counts.items():
        per_file[files.get(f, f)] += c
    for f, c in per_file.most_common():
        print(f"  {f:28s} {c:7d}")
    if want_file:
        rows = sorted(((l, c) for (f, l), c in counts.items() if files.get(f) == want_file))
        # bucket by 10 lines
        for l, c in rows:
            if c >= 40:
                print(f"    {want_file}:{l:5d} {c:6d}")


if __name__ == "__main__":
    main()
