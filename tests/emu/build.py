"""Build tests/emu/libmgx_emu.so (host emulation of the kernel phase functions) with g++."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "libmgx_emu.so")
SOURCES = [os.path.join(HERE, "emu.cpp"), os.path.join(ROOT, "matchering_amd", "csrc", "fir_design.cpp"),
           os.path.join(ROOT, "matchering_amd", "csrc", "fir_plan.cpp")]
HEADERS = [os.path.join(ROOT, "matchering_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "matchering_amd", "csrc"))
           if f.endswith(".h")] + [os.path.join(ROOT, "include", "mgx.h")]


def build(force=False):
    deps = SOURCES + HEADERS
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    cmd = ["g++", "-O2", "-std=c++17", "-DMGX_HOST_EMU", "-shared", "-fPIC", "-ffp-contract=off",
           "-o", OUT] + SOURCES
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
