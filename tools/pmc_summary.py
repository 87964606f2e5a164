"""Summarise rocprofv3 --pmc runs (gpurun_out/pmc_*/.../*_counter_collection.csv) into one JSON:
per kernel, the average of every counter per dispatch and the derived HBM traffic per launch.
Traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of a wide (16 B/lane) read stream, so reads = 2 * FETCH_SIZE * 1024;
WRITE_SIZE is exact for 16-B-per-lane stores.  Narrower accesses are uncalibrated (noted per kernel).
The gather kernels (8-48 B accesses) are upper estimates under that doubling.  When a rocprofv3 --stats kernel csv of an
un-profiled-counter run of the same command is given, its average duration per kernel is added as kernel_ms (the clock a
kernel ran at = SQ_BUSY_CYCLES / 32 shader engines / duration).
usage: python tools/pmc_summary.py "gpurun_out/pmc_b*" profiles/r2_pmc_summary.json [workload] [kernel_stats.csv]
(first argument: glob of the per-pass output directories; csv files are searched below each)"""
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
out["_workload"] = sys.argv[3] if len(sys.argv) > 3 else "C3"
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print("wrote", dst, "kernels:", len(out))
