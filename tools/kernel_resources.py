"""Compile libmgx with -Rpass-analysis=kernel-resource-usage and print one line per kernel."""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matchering_amd import build as b

cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
       "-Rpass-analysis=kernel-resource-usage", "-Wno-unused-value", *[a for a in sys.argv[1:] if a.startswith("-D")], "-o", "/tmp/libmgx_res.so"] + b.SOURCES + ["-L/opt/rocm/lib", "-lrccl", "-lrocprofiler-sdk-roctx"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
want = [a for a in sys.argv[1:] if not a.startswith("-D")] or [""]
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    txt = m.group(1).strip()
    if txt.startswith("Function Name:"):
        cur = {"name": txt.split(":", 1)[1].strip()}
    elif ":" in txt:
        k, v = txt.split(":", 1)
        cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size"):
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            if any(w in name for w in want):
                print(f"{name[:70]:70s} VGPR {cur.get('VGPRs','?'):>4s} AGPR {cur.get('AGPRs','?'):>3s} SGPR {cur.get('TotalSGPRs','?'):>4s} "
                      f"scratch {cur.get('ScratchSize [bytes/lane]','?'):>5s} occ {cur.get('Occupancy [waves/SIMD]','?'):>2s} LDS {cur.get('LDS Size [bytes/block]','?')}")
