"""Build matchering_amd/libmgx.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmgx.so")
SOURCES = [os.path.join(CSRC, "mgx.hip"), os.path.join(CSRC, "fir_design.cpp"), os.path.join(CSRC, "fir_plan.cpp")]


def _deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "mgx.h"))
    return deps


def build(force=False, verbose=False, out=None, extra_flags=()):
    """Compile the HIP library in-tree.  Returns the path of the shared object.

    ``out`` / ``extra_flags`` build an experimental variant next to the product (A/B runs of
    compiler options on the GPU box: ``MGX_LIB=<path>`` makes ``_native`` load it)."""
    out = out or OUT
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in _deps()):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -fno-slp-vectorize: packing the butterflies' float pairs into v_pk_* costs more v_mov shuffles
    # than it saves on gfx950 (k_conv 233 -> 196 us, k_analyze 108 -> 65 us, profiles/r01_d_*)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
           "-fno-slp-vectorize", "-Wno-unused-result", "-Wno-unused-value", *extra_flags, "-o", out] + SOURCES + [
               "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:        # python -m matchering_amd.build --variant NAME FLAG...
        i = sys.argv.index("--variant")
        print(build(force=True, out=os.path.join(HERE, f"libmgx_{sys.argv[i + 1]}.so"), extra_flags=sys.argv[i + 2:]))
    else:
        print(build(force=True, verbose="-v" in sys.argv))
