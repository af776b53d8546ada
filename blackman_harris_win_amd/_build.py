"""In-tree build of the HIP shared library (gfx950 only) and of the test oracle.

`python -m blackman_harris_win_amd._build` or `__graft_entry__.build()`.
hipcc cross-compiles without a GPU; the built .so files stay in-tree (git-ignored) so they
travel to the GPU box with the snapshot.
"""
import hashlib
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


def _digest(paths, extra=""):
    """Content hash of the sources a target is built from: unlike mtimes it survives a copy of the tree (the snapshot
    that goes to the GPU box), so a prebuilt library is rebuilt exactly when a source changed."""
    h = hashlib.sha256(extra.encode())
    for path in paths:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stamp_ok(target, digest):
    try:
        with open(target + ".stamp") as f:
            return os.path.exists(target) and f.read().strip() == digest
    except OSError:
        return False


def _write_stamp(target, digest):
    with open(target + ".stamp", "w") as f:
        f.write(digest + "\n")


KERNEL_UNITS = ("bhw_direct.hip", "bhw_build.hip", "bhw_combine.hip", "bhw_tile9.hip", "bhw_fused.hip", "bhw_taylor.hip", "bhw_variants.hip")
HEADERS = ("bhw_internal.h", "bhw_plan.h", "bhw_device.h", "bhw_tables.inc")


def library_sources():
    srcs = [os.path.join(CSRC, f) for f in ("bhw_api.cpp", "bhw_plan.cpp", "bhw_rom.c") + KERNEL_UNITS + HEADERS]
    srcs.append(os.path.join(ROOT, "include", "bhw.h"))
    return srcs


def compile_units(objdir, flags=(), jobs=None, only=None):
    """hipcc -c of every translation unit (in parallel), gcc -c of the ROM generator; returns (objects, compiler output)."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    # --offload-compress: the gfx950 code objects are stored zstd-compressed in the fat binary (the HIP runtime inflates them at
    # load time): 4.7 MB -> ~1.5 MB of shared library for the same kernels
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "--offload-compress",
            "-Rpass-analysis=kernel-resource-usage"] + list(flags)
    jobsl = []
    for unit in ("bhw_api.cpp",) + KERNEL_UNITS:
        if only and unit not in only:
            continue
        obj = os.path.join(objdir, os.path.splitext(unit)[0] + ".o")
        jobsl.append((obj, base + ["-x", "hip", "-c", os.path.join(CSRC, unit), "-o", obj]))
    rom_o = os.path.join(objdir, "bhw_rom.o")
    plan_o = os.path.join(objdir, "bhw_plan.o")
    if not only:
        jobsl.append((rom_o, ["gcc", "-O2", "-fPIC", "-c", os.path.join(CSRC, "bhw_rom.c"), "-o", rom_o]))
    if not only or "bhw_plan.cpp" in only:
        # the planner is plain C++ by construction (no hip* include): built with g++, like the sanitizer test builds it
        jobsl.append((plan_o, ["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-c", os.path.join(CSRC, "bhw_plan.cpp"), "-o", plan_o]
                      + [f for f in flags if f.startswith("-D")]))
    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
        outs = list(ex.map(lambda j: _run(j[1]), jobsl))
    return [j[0] for j in jobsl], "\n".join(outs)


def link_library(objects, out):
    quad = subprocess.run(["gcc", "-print-file-name=libquadmath.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    _run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + list(objects) +
         [quad, "-Wl,-rpath," + os.path.dirname(os.path.realpath(quad)), "-o", out])


def build_library(force=False, verbose=False):
    extra = os.environ.get("BHW_EXTRA_FLAGS", "")
    digest = _digest(library_sources(), extra)
    if not force and _stamp_ok(LIB, digest):
        return LIB
    objects, out = compile_units(os.path.join(ROOT, "build", "obj"), extra.split())
    link_library(objects, LIB)
    _write_resources(out)
    _write_stamp(LIB, digest)
    if verbose and out:
        print("\n".join(ln for ln in out.splitlines() if "kernel-resource-usage" not in ln and not ln.lstrip().startswith(("|", "^"))
                        and "__global__" not in ln))
    return LIB


RESOURCES = os.path.join(HERE, "kernel_resources.json")


def _write_resources(compiler_output):
    """Per-kernel register / scratch / occupancy figures as the compiler reports them (-Rpass-analysis=kernel-resource-usage),
    kept next to the library (built artefact): tests assert that the headline kernels have no scratch."""
    import json
    import re
    res, cur = {}, None
    for ln in compiler_output.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", ln)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|"
                      r"LDS Size \[bytes/block\]): (\d+)", ln)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    try:
        names = list(res)
        dem = subprocess.run(["c++filt"] + names, stdout=subprocess.PIPE, text=True).stdout.splitlines()
        res = {re.sub(r"^void \(anonymous namespace\)::", "", d).split("(")[0]: res[n] for n, d in zip(names, dem)}
    except OSError:
        pass
    with open(RESOURCES, "w") as f:
        json.dump(res, f, indent=0, sort_keys=True)


def oracle_stale():
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("bhw_oracle.c", "bhw_oracle.h", "cpu_baseline.c", "Makefile")]
    return not _stamp_ok(os.path.join(ORACLE_DIR, "liboracle.so"), _digest(srcs))


def build_oracle(force=False):
    """Compile oracle/liboracle.so (test infrastructure) and, when the upstream checkout is present,
    oracle/_ref from the reference's own cordic() source.  Building the checker is not using it."""
    args = ["make", "-C", ORACLE_DIR, "REF=" + REFERENCE]
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("bhw_oracle.c", "bhw_oracle.h", "cpu_baseline.c", "Makefile")]
    if force:
        _run(["make", "-C", ORACLE_DIR, "clean"])
    else:   # make compares mtimes; a changed source with an older timestamp must still rebuild
        for lib in ("liboracle.so", "libcpubaseline.so"):
            if oracle_stale() and os.path.exists(os.path.join(ORACLE_DIR, lib)):
                os.remove(os.path.join(ORACLE_DIR, lib))
    out = _run(args)
    _write_stamp(os.path.join(ORACLE_DIR, "liboracle.so"), _digest(srcs))
    return out


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
    print(build_oracle())
