"""Build matchering_amd/libmgx.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import hashlib
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


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
         "-Wno-unused-result", "-Wno-unused-value"]


def source_hash(extra_flags=()):
    """Digest of everything the binary is made from.  A content hash, not mtimes: the snapshot that
    carries the tree to a GPU box does not preserve modification order."""
    h = hashlib.sha256(" ".join([*FLAGS, *extra_flags]).encode())
    for path in sorted(_deps()):
        if os.path.isfile(path):
            h.update(os.path.basename(path).encode())
            with open(path, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def _stamp(out):
    return out + ".srchash"


def up_to_date(out=None, extra_flags=()):
    out = out or OUT
    try:
        with open(_stamp(out)) as fh:
            return os.path.exists(out) and fh.read().strip() == source_hash(extra_flags)
    except OSError:
        return False


def build(force=False, verbose=False, out=None, extra_flags=()):
    """Compile the HIP library in-tree.  Returns the path of the shared object.

    ``out`` / ``extra_flags`` build an experimental variant next to the product (A/B runs of
    compiler options on the GPU box: ``MGX_LIB=<path>`` makes ``_native`` load it)."""
    out = out or OUT
    if not force and up_to_date(out, extra_flags):
        return out
    import fcntl
    import tempfile

    # two ranks starting together must not compile onto the same path: one lock per output, the binary is
    # written beside it and moved into place when it is whole
    with open(out + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and up_to_date(out, extra_flags):          # somebody else built it while this process waited
            return out
        fd, partial = tempfile.mkstemp(prefix=os.path.basename(out) + ".", suffix=".part", dir=os.path.dirname(out))
        os.close(fd)
        try:
            _compile(partial, extra_flags, verbose)
            os.chmod(partial, 0o755)
            os.replace(partial, out)
        finally:
            if os.path.exists(partial):
                os.remove(partial)
        with open(_stamp(out), "w") as fh:
            fh.write(source_hash(extra_flags))
    return out


def _compile(out, extra_flags, verbose):
    import shutil

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -fno-slp-vectorize: packing the butterflies' float pairs into v_pk_* costs more v_mov shuffles
    # than it saves on gfx950 (k_conv 233 -> 196 us, k_analyze 108 -> 65 us, profiles/r01_d_*)
    # (the library folder follows the compiler's REAL location: HIPCC=hipcc, a bare name on PATH, must not
    # turn into the relative folder "lib" -- ADVICE round 3)
    found = shutil.which(hipcc)
    rocm_lib = None
    if found:
        rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(found))), "lib")
    if not rocm_lib or not os.path.exists(os.path.join(rocm_lib, "librccl.so")):
        rocm_lib = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")
    roctx = (["-lrocprofiler-sdk-roctx"] if os.path.exists(os.path.join(rocm_lib, "librocprofiler-sdk-roctx.so"))
             else ["-DMGX_NO_ROCTX"])             # (ROCm installs without the profiler SDK: the stage markers go)
    cmd = [hipcc, *FLAGS, *extra_flags, *roctx, "-o", out] + SOURCES + [f"-L{rocm_lib}", "-lrccl", f"-Wl,-rpath,{rocm_lib}"]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)


if __name__ == "__main__":
    if "--variant" in sys.argv:        # python -m matchering_amd.build --variant NAME FLAG...  -> tools/variants/libmgx_NAME.so
        i = sys.argv.index("--variant")
        folder = os.path.join(os.path.dirname(HERE), "tools", "variants")      # experiments stay out of the product package
        os.makedirs(folder, exist_ok=True)
        print(build(force=True, out=os.path.join(folder, f"libmgx_{sys.argv[i + 1]}.so"), extra_flags=sys.argv[i + 2:]))
    else:
        print(build(force=True, verbose="-v" in sys.argv))
