"""Where a step's time goes on the GPU timeline, from a rocprofv3 --kernel-trace csv (*_kernel_trace.csv).

    python tools/timeline.py gpurun_out/<tag>/trace/*/*_kernel_trace.csv [last_n_steps]

A step starts at each gsr_preprocess_kernel dispatch.  For the last N complete steps prints, per dispatch position in the
step, the median start offset, duration and the idle gap in front of it (start minus the latest end of everything
dispatched before it in the step -- kernels on the helper stream overlap, so the gap is taken against the running
maximum of the ends), then the sums: kernel time on the critical path, idle time, step length."""
import csv
import statistics
import sys

args = [x for x in sys.argv[1:] if x not in ("--all", "--raw")]
every = "--all" in sys.argv   # every kernel of the process (a training iteration: PyTorch's kernels between the library's)
raw = "--raw" in sys.argv     # the last N dispatches as they are, with start / end relative to the first: overlapping streams (views in flight)
path = args[0]
last = int(args[1]) if len(args) > 1 else 10
rows = []
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if every or name.startswith("gsr_") or "rocclr" in name:   # the library's kernels and the runtime's copy / fill kernels between them
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
if raw:
    sel = rows[-last:]
    t0 = sel[0][0]
    print(f"{'kernel':44s} {'start us':>9} {'end us':>9} {'dur us':>8}  (the last {len(sel)} dispatches of the run, by start time; kernels of different streams overlap)")
    for a, b, name in sel:
        print(f"{name[:44]:44s} {(a - t0) / 1e3:9.1f} {(b - t0) / 1e3:9.1f} {(b - a) / 1e3:8.1f}")
    sys.exit(0)
starts = [i for i, r in enumerate(rows) if r[2].startswith("gsr_preprocess_kernel")]
steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
# keep the steps of the most common length (the timed ones; per-kernel-table steps run the colour kernel in line)
steps = [[r for r in s if not ((r[2].startswith("__amd_rocclr_fill") or r[2].startswith("gsr_zero_status")) and r is s[-1])] for s in steps]   # the next step's status zeroing
if not every:
    steps = [s for s in steps if s and s[-1][2].startswith("gsr_gaussian_backward")]
n = statistics.mode(len(s) for s in steps)
steps = [s for s in steps if len(s) == n]
# the timed steps run the colour kernel on the helper stream beside the geometry kernel; the steps of bench.py's per-kernel
# table run it in line and drain the queue around every stage (event records): prefer the former
beside = [s for s in steps if any(r[2].startswith("gsr_preprocess_color") and r[0] < s[0][1] for r in s[1:3])]
steps = (beside or steps)[-last:]
print(f"{len(steps)} steps of {n} dispatches each")
print(f"{'#':>2} {'kernel':44s} {'start us':>9} {'dur us':>8} {'gap us':>7}")
tot_gap = tot_busy = 0.0
for k in range(n):
    st, du, gp = [], [], []
    for s in steps:
        t0 = s[0][0]
        end_before = max([e for (_, e, _) in s[:k]], default=t0)
        st.append((s[k][0] - t0) / 1e3)
        du.append((s[k][1] - s[k][0]) / 1e3)
        gp.append((s[k][0] - end_before) / 1e3)
    g = statistics.median(gp)
    print(f"{k:2d} {steps[0][k][2][:44]:44s} {statistics.median(st):9.1f} {statistics.median(du):8.1f} {g:7.1f}")
    tot_gap += max(g, 0.0)
lens = [(max(e for _, e, _ in s) - s[0][0]) / 1e3 for s in steps]
period = [(b[0][0] - a[0][0]) / 1e3 for a, b in zip(steps[:-1], steps[1:]) if b[0][0] - a[0][0] < 3 * (a[-1][1] - a[0][0])]
print(f"first start -> last end: median {statistics.median(lens):.1f} us; idle gaps inside: {tot_gap:.1f} us; "
      f"step period (start to next start): median {statistics.median(period) if period else float('nan'):.1f} us")
