#!/bin/bash
# Calibrates the SQ counters the issue-fraction metric is built from against kernels of KNOWN instruction counts:
# tools/valu_probe runs, per instruction kind and 1 / 4 / 8 waves per SIMD, nwaves x 2048 x 64 instructions of that kind
# (plus a few dozen of set-up).  usage (GPU box, repo root): bash tools/pmc_calib.sh <outdir>
set -u
OUT=${1:?outdir}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_TRANS_F32 --output-format csv -d "$OUT/pmc" -- "$ROOT/tools/valu_probe" > "$OUT/probe_under_pmc.txt" 2>&1 || tail -3 "$OUT/probe_under_pmc.txt"
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_grbm" -- "$ROOT/tools/valu_probe" > "$OUT/probe_under_grbm.txt" 2>&1 || tail -3 "$OUT/probe_under_grbm.txt"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --output-format csv -d "$OUT/pmc_class" -- "$ROOT/tools/valu_probe" > "$OUT/probe_under_class.txt" 2>&1 || tail -3 "$OUT/probe_under_class.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(dict)
for f in glob.glob(out + "/pmc*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"].split("(")[0], int(r["Dispatch_Id"]))
        rows[key][r["Counter_Name"]] = float(r["Counter_Value"])
        rows[key]["_dur_" + ("grbm" if "grbm" in f else "sq")] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        rows[key]["_wg"] = int(r.get("Workgroup_Size", 0) or 0)
        rows[key]["_grid"] = int(r.get("Grid_Size", 0) or 0)
with open(out + "/calibration.txt", "w") as o:
    print("kernel dispatch grid wg | known VALU instr | SQ_INSTS_VALU ratio | ACTIVE_INST_VALU/INSTS | BUSY_CYCLES/32/dur(ns) GHz | 4*WAVE_CYCLES/waves/dur GHz-equivalent | GRBM/8/dur GHz", file=o)
    for (k, d), c in sorted(rows.items(), key=lambda x: x[0][1]):
        if "SQ_INSTS_VALU" not in c and "GRBM_GUI_ACTIVE" not in c:
            continue
        waves = c["_grid"] // 64 if c["_grid"] else 0
        known = waves * 2048 * 64
        line = f"{k[:28]:28s} {d:4d} {c['_grid']:8d} {c['_wg']:5d} | {known:12d} |"
        if "SQ_INSTS_VALU" in c:
            line += f" {c['SQ_INSTS_VALU'] / max(known, 1):6.3f} | {c.get('SQ_ACTIVE_INST_VALU', 0) / max(c['SQ_INSTS_VALU'], 1):6.3f} | {c.get('SQ_BUSY_CYCLES', 0) / 32 / c['_dur_sq']:6.3f} | {4 * c.get('SQ_WAVE_CYCLES', 0) / max(waves, 1) / c['_dur_sq']:6.3f} |"
        if "GRBM_GUI_ACTIVE" in c:
            line += f" grbm {c['GRBM_GUI_ACTIVE'] / 8 / c['_dur_grbm']:6.3f}"
        if "SQ_INSTS_VALU_MUL_F32" in c:   # which class counter an instruction kind lands in, per known instruction
            line += " | class/known: " + " ".join(f"{n[14:]} {c.get(n, 0) / max(known, 1):.2f}" for n in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT"))
        print(line, file=o)
print(open(out + "/calibration.txt").read())
PY
