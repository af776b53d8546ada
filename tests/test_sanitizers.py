"""AddressSanitizer + UBSan on the CPU build (GPU sanitizers are not available on the pool): the oracle's restatement, the
product's host-side ROM derivation, and the product's whole HIP-free planning unit (bhw_plan.cpp: validation, resolution into
kernel constants, strategy / format / tile-plan / ownership / scratch arithmetic, bhw_describe_plan) are swept over all
models / rules / widths / part counts with no report."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_rom_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "san_oracle")
    subprocess.run(["gcc", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-I" + os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "c", "san_oracle.c"),
                    os.path.join(ROOT, "oracle", "bhw_oracle.c"),
                    os.path.join(ROOT, "blackman_harris_win_amd", "csrc", "bhw_rom.c"),
                    "-lquadmath", "-lm", "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.startswith("ok ") and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_planner_clean_under_asan_ubsan_over_the_parameter_lattice(tmp_path):
    """SURVEY section 5 / round-3 verdict: the host side of the product, not only the oracle, runs under the sanitizers.  bhw_plan.cpp
    holds every decision the library takes from parameters alone and includes no HIP header, so it builds with plain g++;
    tests/cpp/san_plan.cpp walks every (model, rule, source, window, phi_width 3..31, dat_width 7..33, precision, lut_size) --
    valid or not --, every call shape / strategy / table-format limit / verdict state, the tile plans, the fused kernel's forms and
    the ownership segments for 1..64 parts, and checks the invariants the launch code relies on (quarter circle within 2^32 for the
    32-bit kernels, scratch within the documented bound, parts covering the window)."""
    exe = str(tmp_path / "san_plan")
    csrc = os.path.join(ROOT, "blackman_harris_win_amd", "csrc")
    src = open(os.path.join(csrc, "bhw_plan.cpp")).read() + open(os.path.join(csrc, "bhw_plan.h")).read()
    assert "hip/" not in src and "hipError" not in src and "hipStream" not in src      # HIP-free by construction
    subprocess.run(["g++", "-g", "-O2", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I" + csrc,
                    os.path.join(ROOT, "tests", "cpp", "san_plan.cpp"), os.path.join(csrc, "bhw_plan.cpp"), "-o", exe],
                   check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    assert r.stdout.startswith("ok ") and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
    assert int(r.stdout.split()[1]) > 50000
