"""Writes the error-versus-bar lines of the parity tests to a file (pytest -q drops what the tests print).

    python tools/parity_report.py gpurun_out/r3_parity_report.txt [c1 C2 C3 C5 C4]

Runs the SAME test functions pytest runs (tests/test_parity_gpu.py, tests/test_fullsize_gpu.py) in this process on
cuda:0 and keeps what they log through tests/util.py parity_log(): per case the instance count, the image error on the
non-fragile pixels, every gradient tensor's error next to its bar at the three levels of check_grads (blend sums, chain,
end to end), the exclusion figures (fragile-pixel fraction, image error over all pixels, gradients with an unmasked
upstream gradient) and the flip attribution.  A failing case is recorded with its message and the run goes on."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting_cc-comments_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import util  # noqa: E402
import test_fullsize_gpu as full  # noqa: E402
import test_parity_gpu as par  # noqa: E402


def main():
    out = sys.argv[1]
    want = [a.lower() for a in sys.argv[2:]] or ["c1", "c2", "c3", "c5", "c4"]
    import tempfile
    cases = []
    if "c1" in want:
        for c in par.CASES:
            cases.append((c[0], lambda c=c: par.test_forward_backward_vs_oracle(*c)))
    if "c2" in want:
        cases.append(("C2", full.test_c2_full_parity_with_oracle))
    if "c3" in want:
        cases.append(("C3", full.test_c3_full_parity_with_oracle))
    if "c5" in want:
        cases.append(("C5", full.test_c5_full_parity_with_oracle))
    if "c4" in want:
        cases.append(("C4", lambda: full.test_c4_eight_views_summed_gradients_match_oracle(tempfile.mkdtemp())))
    from diff_gaussian_rasterization import _C
    head = [f"parity report: HIP path ({_C.lib().gsr_version().decode()}, {torch.cuda.get_device_name(0)}) against the CPU oracle "
            "(oracle/gsr_oracle.c), same seeded inputs",
            "bars: integer outputs exact; image 1e-5 absolute on non-fragile pixels; gradients max(1e-5, 2 x the reference's own "
            "fp32-atomic reproducibility band), max-abs error / max|gradient| per tensor", ""]
    with open(out, "w") as f:
        f.write("\n".join(head) + "\n")
        for name, fn in cases:
            util.PARITY_LOG.clear()
            t0 = time.time()
            try:
                fn()
                status = "PASS"
            except Exception:  # noqa: BLE001
                status = "FAIL\n" + traceback.format_exc()
            f.write(f"==== {name}: {status} ({time.time() - t0:.0f} s)\n" + "\n".join(util.PARITY_LOG) + "\n\n")
            f.flush()
            print(f"{name}: {status.splitlines()[0]} ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
