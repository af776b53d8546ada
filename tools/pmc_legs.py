#!/usr/bin/env python3
"""A few launches of every extra leg of bench.py (and of the headline step), for the hardware-counter passes of tools/gpu_pmc_legs.sh:
each leg's kernels are identified by name, so the legs run one after the other in one process and the counters are averaged per
kernel name and launch.  Prints one JSON line: leg -> {plan, kernels (name prefixes as rocprofv3 prints them), coefficients, bytes}."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import blackman_harris_win_amd as bhw  # noqa: E402
from blackman_harris_win_amd import binding as B  # noqa: E402

N26 = 1 << 26
REPS = int(os.environ.get("PMC_REPS", "6"))


def kernels_of(plan, extra=()):
    """kernel-name prefixes of a bhw_describe_plan line, in rocprofv3's spelling (a blank after every comma)"""
    body = plan.rsplit(": ", 1)[1]                  # ("bhw_apply_device: table[nibble]: k_a + k_b" -> "k_a + k_b")
    names = [s.strip().split(" ")[0] for s in body.split(" + ")]
    names = [n for n in names if n.startswith("k_")] + list(extra)
    return [n.replace(",", ", ").rstrip(">") for n in names]


LEGS = ("headline_C3", "C2_bh4_2^20_24bit", "C4_1024x_bh4_2^16_24bit", "bh7_2^26_16bit_cpp", "taylor_hamming_2^26_16bit", "C3_vhdl_cosine_sum",
        "C3_vhdl_cordic_and_sum", "C3_model_cpp", "fused_apply_C3", "bh7_2^16_32bit")


def main(which):
    """One leg per process (several legs launch kernels of the same name: C2 and the first frame of C4, the fused apply and the
    headline window): REPS calls after one settling call; the plan line is read AFTER the calls, when the table format is settled."""
    out = torch.empty(N26, dtype=torch.int32, device="cuda")

    def leg(fn, plan, n_coeff, bytes_per=4, extra=()):
        fn()
        torch.cuda.synchronize()
        for _ in range(REPS):
            fn()
        torch.cuda.synchronize()
        plan = plan()
        print(json.dumps({which: {"plan": plan, "kernels": kernels_of(plan, extra), "coefficients": n_coeff, "algorithmic_bytes": bytes_per * n_coeff,
                                  "launches": REPS + 1}}), flush=True)

    if which == "headline_C3":
        p = bhw.make_params(7, 26, 32)
        leg(lambda: bhw.generate(p, 0, N26, out=out), lambda: B.describe_plan(p, 0, N26), N26)
    elif which == "C2_bh4_2^20_24bit":
        p = bhw.make_params(4, 20, 24)
        leg(lambda: bhw.generate(p, 0, 1 << 20, out=out[:1 << 20]), lambda: B.describe_plan(p, 0, 1 << 20), 1 << 20)
    elif which == "C4_1024x_bh4_2^16_24bit":
        p = bhw.make_params(4, 16, 24)
        o4 = out.view(1024, 1 << 16)
        leg(lambda: bhw.generate_batched(p, 1024, out=o4), lambda: B.describe_plan(p, 0, 1 << 16), N26)      # (one launch computes the period and writes all 1024 frames)
    elif which == "bh7_2^26_16bit_cpp":
        p = bhw.make_params(7, 26, 16, model=B.MODEL_CPP)
        leg(lambda: bhw.generate(p, 0, N26, out=out), lambda: B.describe_plan(p, 0, N26), N26)
    elif which == "taylor_hamming_2^26_16bit":
        p = bhw.make_params(1, 26, 16, sin_type=B.SIN_TAYLOR, combine=B.COMBINE_VHDL, lut_size=9)
        leg(lambda: bhw.generate(p, 0, N26, out=out), lambda: "taylor: k_taylor_window_fold", N26)
    elif which == "C3_vhdl_cosine_sum":
        p = bhw.make_params(7, 26, 32, combine=B.COMBINE_VHDL)
        leg(lambda: bhw.generate(p, 0, N26, out=out), lambda: B.describe_plan(p, 0, N26), N26)
    elif which == "C3_vhdl_cordic_and_sum":
        p = bhw.make_params(7, 26, 32, model=B.MODEL_VHDL, combine=B.COMBINE_VHDL)
        leg(lambda: bhw.generate(p, 0, N26, out=out), lambda: B.describe_plan(p, 0, N26), N26)
    elif which == "C3_model_cpp":
        p = bhw.make_params(7, 26, 32, model=B.MODEL_CPP)
        leg(lambda: bhw.generate(p, 0, N26, out=out), lambda: B.describe_plan(p, 0, N26), N26)
    elif which == "bh7_2^16_32bit":
        p = bhw.make_params(7, 16, 32)
        leg(lambda: bhw.generate(p, 0, 1 << 16, out=out[:1 << 16]), lambda: B.describe_plan(p, 0, 1 << 16), 1 << 16)
    elif which == "fused_apply_C3":
        x = torch.randint(-(1 << 31), (1 << 31) - 1, (N26,), dtype=torch.int32, device="cuda")
        p = bhw.make_params(7, 26, 32)
        leg(lambda: bhw.apply(p, x, out=out, shift=31), lambda: "bhw_apply_device: " + B.describe_plan(p, 0, N26), N26, bytes_per=8)
    else:
        raise SystemExit("unknown leg " + which)


if __name__ == "__main__":
    main(sys.argv[1])
