"""In-tree build of the HIP shared library (gfx950 only) and of the test oracle.

`python -m blackman_harris_win_amd._build` or `__graft_entry__.build()`.
hipcc cross-compiles without a GPU; the built .so files stay in-tree (git-ignored) so they
travel to the GPU box with the snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbhw.so")
ORACLE_DIR = os.path.join(ROOT, "oracle")
REFERENCE = "/root/reference"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, cwd=None):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build step failed: " + " ".join(cmd))
    return r.stdout


def build_library(force=False, verbose=False):
    srcs = [os.path.join(CSRC, f) for f in ("bhw_api.cpp", "bhw_kernels.hip", "bhw_rom.c", "bhw_internal.h", "bhw_tables.inc")]
    srcs.append(os.path.join(ROOT, "include", "bhw.h"))
    if not force and not _newer(LIB, srcs):
        return LIB
    rom_o = os.path.join(CSRC, "bhw_rom.o")
    _run(["gcc", "-O2", "-fPIC", "-c", os.path.join(CSRC, "bhw_rom.c"), "-o", rom_o])
    quad = subprocess.run(["gcc", "-print-file-name=libquadmath.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    extra = os.environ.get("BHW_EXTRA_FLAGS", "").split()
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function"] + extra + ["-x", "hip",
           os.path.join(CSRC, "bhw_api.cpp"), os.path.join(CSRC, "bhw_kernels.hip"),
           "-x", "none", rom_o, quad, "-Wl,-rpath," + os.path.dirname(os.path.realpath(quad)),
           "-o", LIB]
    out = _run(cmd)
    if verbose and out:
        print(out)
    return LIB


def build_oracle(force=False):
    """Compile oracle/liboracle.so (test infrastructure) and, when the upstream checkout is present,
    oracle/_ref from the reference's own cordic() source.  Building the checker is not using it."""
    args = ["make", "-C", ORACLE_DIR, "REF=" + REFERENCE]
    if force:
        _run(["make", "-C", ORACLE_DIR, "clean"])
    return _run(args)


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
    print(build_oracle())
