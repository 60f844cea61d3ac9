"""Summarise rocprofv3 --pmc csv output: per (kernel, grid size), mean counter value per dispatch.
usage: python tools/pmc_summary.py gpurun_out/<tag> profiles/<name>.json [dir_prefix]
dir_prefix (default "pmc_") selects the pass directories ("prv_" = the prover-round passes).

Dispatches of one kernel template with different grids (batch 1 / batch 4 / another size) are kept
apart, so an entry is one measured configuration.  HBM bytes are reported RAW (FETCH_SIZE + WRITE_SIZE,
KiB -> bytes) and CORRECTED (2 x FETCH_SIZE + WRITE_SIZE): on gfx950 FETCH_SIZE tallies a 128-byte
request of a wide coalesced streaming read at 64 bytes (MI355X_MICROARCH.md, HBM); the factor is
calibrated for 16-byte-per-lane streaming reads only -- for the random 96-byte gathers of
msm_accumulate_l1 the truth lies between the two numbers.  Nothing in bench.py reads this file: it is
offline evidence kept under profiles/."""
import csv, glob, json, os, sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
prefix = sys.argv[3] if len(sys.argv) > 3 else "pmc_"
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, prefix + "*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name") or row.get("Kernel Name")
            grid = row.get("Grid_Size") or row.get("Grid Size") or "?"
            wg = row.get("Workgroup_Size") or row.get("Workgroup Size") or "?"
            acc[(name, grid, wg)][row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {}
for (kname, grid, wg), ctrs in acc.items():
    if not kname.startswith(("void pm::", "pm::")):
        continue
    short = kname.split("(")[0].replace("void ", "")
    s = {c: {"mean": sum(v) / len(v), "dispatches": len(v)} for c, v in ctrs.items()}
    if "FETCH_SIZE" in s and "WRITE_SIZE" in s:
        s["hbm_bytes_per_launch_raw"] = (s["FETCH_SIZE"]["mean"] + s["WRITE_SIZE"]["mean"]) * 1024
        s["hbm_bytes_per_launch_corrected"] = (2 * s["FETCH_SIZE"]["mean"] + s["WRITE_SIZE"]["mean"]) * 1024
    if "SQ_INSTS_VALU" in s and "SQ_BUSY_CYCLES" in s:
        pass
    summary[f"{short} grid={grid} wg={wg}"] = s
json.dump(summary, open(out, "w"), indent=1, sort_keys=True)
for k, v in sorted(summary.items()):
    print(k, {c: (round(x["mean"]) if isinstance(x, dict) else round(x)) for c, x in v.items()})
