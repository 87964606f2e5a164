"""Summarise rocprofv3 --pmc runs (gpurun_out/pmc_*/.../*_counter_collection.csv) into one JSON:
per kernel, the average of every counter per dispatch and the derived HBM traffic per launch.
Traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of a wide (16 B/lane) read stream, so reads = 2 * FETCH_SIZE * 1024;
WRITE_SIZE is exact for 16-B-per-lane stores.  Narrower accesses are uncalibrated (noted per kernel).
usage: python tools/pmc_summary.py "gpurun_out/pmc_b*" profiles/r1_pmc_summary.json [workload]
(first argument: glob of the per-pass output directories; csv files are searched below each)"""
import collections
import csv
import glob
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
files = [f for d in glob.glob(src) for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)]
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("gsr_"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in sorted(agg.items()):
    if not c:
        continue
    e = {n: sum(v) / len(v) for n, v in c.items()}
    e["dispatches_sampled"] = max(len(v) for v in c.values())
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_read_bytes"] = 2.0 * e["FETCH_SIZE"] * 1024.0
        e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024.0
        e["hbm_traffic_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
    if "SQ_ACTIVE_INST_VALU" in e and "SQ_WAVE_CYCLES" in e:
        e["valu_active_over_wave_cycles"] = e["SQ_ACTIVE_INST_VALU"] / e["SQ_WAVE_CYCLES"]
    out[k] = e
out["_workload"] = sys.argv[3] if len(sys.argv) > 3 else "C3"
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print("wrote", dst, "kernels:", len(out))
