"""The table-build kernel's exact short cuts, checked on the host (no GPU): tools/sim_build.cpp is a CPU model of
k_table_build_mirror's arithmetic -- octant mirror with zero-event deferral, narrow 32-bit state behind the scalar group test (and
its biased form, x - 2^(NITER-1), for the groups next to 0 degrees), table-driven tail behind the margin test -- and compares every entry of the first-quadrant table with the plain rotation loop
(hls/windows/win_function.cpp:110-125 | cpp/cordic_sincos.cpp:49-63 | src/cordic_dds.vhd:197-213)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("sim") / "sim_build")
    subprocess.check_call(["g++", "-O2", "-o", exe, os.path.join(ROOT, "tools", "sim_build.cpp"), "-lquadmath"])
    return exe


# (model, PW, W, precision): the three bit-models at the headline width, narrower widths (more lanes outside the tail margins),
# and a PRECISION-2 VHDL configuration (amplitude beyond 2^NITER: every group takes the wide path)
@pytest.mark.parametrize("cfg", [(0, 22, 32, 1), (1, 22, 32, 1), (2, 22, 32, 1), (0, 24, 30, 1), (0, 22, 24, 1), (1, 23, 24, 1),
                                 (2, 22, 23, 1), (2, 22, 30, 2)])
def test_every_table_entry_matches_the_plain_chain(sim, cfg):
    r = subprocess.run([sim] + [str(v) for v in cfg], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches 0" in r.stdout
    if cfg[3] == 1 and cfg[2] == 32:
        # the short cuts are actually taken: almost every group is narrow, almost no wave leaves the tail tables
        groups = int(r.stdout.split("groups ")[1].split(",")[0])
        wide = int(r.stdout.split("wide-state groups ")[1].split(" ")[0])
        unsafe = int(r.stdout.split("unsafe tail lane ")[1].split(" ")[0])
        assert wide < 0.04 * groups and unsafe < 0.01 * groups


def test_headline_table_2_26_32(sim):
    """All 2^24 entries of the BASELINE configs[2] table (HLS model, 2^26 / 32 bits): 2.5 s on one core."""
    r = subprocess.run([sim, "0", "26", "32"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "mismatches 0" in r.stdout, r.stdout + r.stderr
    # round 4: the groups whose x reaches 2^32 (the first third of a degree) run the biased narrow state, not the 64-bit loop:
    # 659 of them here, and what is left on the 64-bit path (groups that split before the ten-rotation block) is a quarter per cent
    biased = int(r.stdout.split("biased-narrow groups ")[1].split(",")[0])
    wide = int(r.stdout.split("wide-state groups ")[1].split(" ")[0])
    assert biased > 500 and wide < 400
