"""Summarise rocprofv3 --pmc runs (gpurun_out/pmc_*/.../*_counter_collection.csv) into one JSON:
per kernel, the average of every counter per dispatch and the derived HBM traffic per launch.
Traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of a wide (16 B/lane) read stream, so reads = 2 * FETCH_SIZE * 1024;
WRITE_SIZE is exact for 16-B-per-lane stores.  Narrower accesses are uncalibrated (noted per kernel).
The gather kernels (8-48 B accesses) are upper estimates under that doubling.  When a rocprofv3 --stats kernel csv of an
un-profiled-counter run of the same command is given, its average duration per kernel is added as kernel_ms (the clock a
kernel ran at = SQ_BUSY_CYCLES / 32 shader engines / duration).
usage: python tools/pmc_summary.py "gpurun_out/pmc_b*" profiles/r2_pmc_summary.json [workload] [kernel_stats.csv]
(first argument: glob of the per-pass output directories; csv files are searched below each)

Vector-ALU roof (round 3, replaces the flat "4 cycles per instruction, 8 per transcendental" of round 2, which read 1.17 for
the forward blend).  tools/valu_probe.hip measures what the chip sustains per instruction KIND with 4-8 waves per SIMD, in
instructions per second over all 1024 SIMDs -- a time-domain figure, so no clock estimate enters (profiles/r3_valu_probe.txt,
w = 5 rows): v_add_f32 / v_mul_f32 (VOP2) 904 / 852 G/s, v_fma / v_fmac 630 / 617, v_cmp / v_cndmask / v_max / DPP adds
505-576, the packed v_pk_add / v_pk_mul / v_pk_fma 534 / 481 / 526 (two pixels each), v_exp / v_rcp 300, the lane swaps 276.
tools/pmc_calib.sh checked against those known-count kernels that SQ_INSTS_VALU counts instructions exactly and that
SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32 count one per instruction, packed or not (profiles/r3_pmc_calibration.txt).  Per kernel:
    valu_roof_ms  = sum over classes of count x cost(class) / 1024 SIMDs,  cost = 1 / rate
    valu_roof_frac = valu_roof_ms / kernel_ms     (kernel_ms of the un-profiled run)
with the class counts from the counters and "other" = SQ_INSTS_VALU minus the four classes (compares, selects, moves, min/max,
cross-lane moves: 1.85 ns).  The counters cannot tell a packed instruction from a plain one, so the share of packed (and, for
adds, DPP) instructions per class is a per-kernel constant read off the ISA of its loops (VALU_MIX below).

That additive model is PESSIMISTIC for a mixed stream: waves with cheap and expensive instructions overlap better than the sum of
their single-kind costs (at C5, where the forward blend has no tail to speak of, it read 1.12).  For the two blend kernels the roof
is therefore measured, not composed: the probe runs a dependency-free stream with the SAME class mix as the kernel (kinds
"forward / backward blend class mix": 22 resp. 25 instructions in the proportions the class counters report) and the chip's
sustained rate on it, at the kernel's occupancy, is the roof:
    valu_roof_ms = SQ_INSTS_VALU / MIX_RATE[kernel],   forward 779 G instr/s (6 waves per SIMD), backward 588 (4 waves per SIMD)
(profiles/r3_valu_probe.txt).  The additive figure is kept beside it as valu_roof_ms_additive."""
import collections
import csv
import glob
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> dispatch durations (ns) IN THAT PASS
files = [f for d in glob.glob(src) for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)]
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("gsr_"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                dur[k][r["Counter_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {}
for k, c in sorted(agg.items()):
    if not c:
        continue
    e = {n: sum(v) / len(v) for n, v in c.items()}
    e["dispatches_sampled"] = max(len(v) for v in c.values())
    if dur[k].get("SQ_BUSY_CYCLES"):
        e["sq_pass_kernel_ms"] = sum(dur[k]["SQ_BUSY_CYCLES"]) / len(dur[k]["SQ_BUSY_CYCLES"]) * 1e-6   # duration while the SQ counters ran
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_read_bytes"] = 2.0 * e["FETCH_SIZE"] * 1024.0
        e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024.0
        e["hbm_traffic_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
    if "SQ_ACTIVE_INST_VALU" in e and "SQ_WAVE_CYCLES" in e:
        e["valu_active_over_wave_cycles"] = e["SQ_ACTIVE_INST_VALU"] / e["SQ_WAVE_CYCLES"]
    out[k] = e
if len(sys.argv) > 4:
    for r in csv.DictReader(open(sys.argv[4])):
        k = r["Name"].split("(")[0].replace("void ", "")
        if k in out:
            out[k]["kernel_ms"] = float(r["AverageNs"]) * 1e-6
            out[k]["kernel_calls"] = int(r["Calls"])
for k, e in out.items():
    if "sq_pass_kernel_ms" in e and "SQ_BUSY_CYCLES" in e:
        # SQ_BUSY_CYCLES is summed over the 32 shader engines; duration of the SAME pass (counter collection slows a kernel)
        e["clock_GHz"] = e["SQ_BUSY_CYCLES"] / 32.0 / (e["sq_pass_kernel_ms"] * 1e-3) / 1e9
    if "SQ_INSTS_VALU" in e and "SQ_BUSY_CYCLES" in e:
        trans = e.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
        # issue cycles: 4 per wave64 VALU instruction, 8 per transcendental (tools/valu_probe.hip), 1024 SIMDs;
        # available: the kernel's own cycles = SQ_BUSY_CYCLES / 32 per SIMD
        e["valu_issue_frac"] = (4.0 * (e["SQ_INSTS_VALU"] - trans) + 8.0 * trans) / (1024.0 * e["SQ_BUSY_CYCLES"] / 32.0)
    if "SQ_WAVE_CYCLES" in e and "SQ_BUSY_CYCLES" in e:
        # SQ_WAVE_CYCLES counts quad-cycles summed over waves: average waves resident per SIMD during the kernel
        e["avg_waves_per_simd"] = 4.0 * e["SQ_WAVE_CYCLES"] / 1024.0 / (e["SQ_BUSY_CYCLES"] / 32.0)
# cost in ns of one wave64 instruction on one SIMD = 1024 / (chip rate in G instr/s), profiles/r3_valu_probe.txt (w = 5)
COST = dict(add=1024 / 904.0, add_dpp=1024 / 555.0, pk_add=1024 / 534.0, mul=1024 / 852.0, pk_mul=1024 / 481.0, fma=1024 / 623.0,
            pk_fma=1024 / 526.0, trans=1024 / 300.0, other=1024 / 553.0)
# share of each class's instructions that are packed (v_pk_*_f32) / DPP adds, from the ISA of the kernel's loops weighted by the
# per-tile work counters of tools/tile_clock.py (profiles/r3_tile_clock_c3_uniform.txt): the backward blend evaluates its pixel
# pairs with packed instructions and reduces with plain and DPP adds; nothing else on the path uses packed arithmetic
VALU_MIX = {"gsr_render_backward_wave_kernel": dict(pk_add=0.39, dpp_add=0.21, pk_mul=0.88, pk_fma=0.93)}
# chip-wide sustained rate (G wave64 instructions / s) of the probe's class-mix streams at the kernels' occupancies
MIX_RATE = {"gsr_render_forward_wave_kernel": 779.0, "gsr_render_backward_wave_kernel": 588.0}
for k, e in out.items():
    if all(n in e for n in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32")) and "kernel_ms" in e:
        mix = VALU_MIX.get(k.split("<")[0], {})
        a, m, f = e["SQ_INSTS_VALU_ADD_F32"], e["SQ_INSTS_VALU_MUL_F32"], e["SQ_INSTS_VALU_FMA_F32"]
        t = e.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
        other = max(0.0, e["SQ_INSTS_VALU"] - a - m - f - t)
        pa, da, pm, pf = mix.get("pk_add", 0.0), mix.get("dpp_add", 0.0), mix.get("pk_mul", 0.0), mix.get("pk_fma", 0.0)
        ns = (a * (pa * COST["pk_add"] + da * COST["add_dpp"] + (1 - pa - da) * COST["add"]) + m * (pm * COST["pk_mul"] + (1 - pm) * COST["mul"]) +
              f * (pf * COST["pk_fma"] + (1 - pf) * COST["fma"]) + t * COST["trans"] + other * COST["other"])
        e["valu_roof_ms_additive"] = ns / 1024.0 * 1e-6
        rate = MIX_RATE.get(k.split("<")[0])
        e["valu_roof_ms"] = e["SQ_INSTS_VALU"] / (rate * 1e9) * 1e3 if rate else e["valu_roof_ms_additive"]
        e["valu_roof_model"] = "measured rate of a synthetic stream of the kernel's class mix" if rate else "additive per-class costs"
        e["valu_roof_frac"] = e["valu_roof_ms"] / e["kernel_ms"]
        e["valu_class_counts"] = dict(add=a, mul=m, fma=f, trans=t, other=other)
out["_workload"] = sys.argv[3] if len(sys.argv) > 3 else "C3"
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print("wrote", dst, "kernels:", len(out))
