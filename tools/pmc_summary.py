"""Summarise rocprofv3 --pmc csv output: per kernel, mean counter value per dispatch.
usage: python tools/pmc_summary.py gpurun_out/r01 profiles/r01_pmc_summary.json [log_n [dir_prefix]]
dir_prefix (default "pmc_") selects the pass directories; "prv_" = the prover-round passes, for which
only the summary json is written (no ntt_traffic.json)."""
import csv, glob, json, os, sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
prefix = sys.argv[4] if len(sys.argv) > 4 else "pmc_"
for path in glob.glob(os.path.join(root, prefix + "*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name") or row.get("Kernel Name")
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {}
for kname, ctrs in acc.items():
    if not kname.startswith(("void pm::", "pm::")):
        continue
    short = kname.split("(")[0].replace("void ", "")
    summary[short] = {c: {"mean": sum(v) / len(v), "dispatches": len(v)} for c, v in ctrs.items()}
    s = summary[short]
    if "FETCH_SIZE" in s and "WRITE_SIZE" in s:
        # gfx950: FETCH_SIZE (KiB) counts 128-B requests at 64 B for wide streaming reads -> x2 (MI355X_MICROARCH.md, HBM)
        s["hbm_bytes_per_launch_corrected"] = (2 * s["FETCH_SIZE"]["mean"] + s["WRITE_SIZE"]["mean"]) * 1024
        s["hbm_bytes_per_launch_raw"] = (s["FETCH_SIZE"]["mean"] + s["WRITE_SIZE"]["mean"]) * 1024
json.dump(summary, open(out, "w"), indent=1, sort_keys=True)
# bench.py reads profiles/ntt_traffic.json: {"<role>_2^<k>": corrected HBM bytes per launch}
import re
traffic = {}
roles = {("true", "false", "true"): "ntt_pass_first", ("false", "true", "true"): "ntt_pass_middle",
         ("false", "true", "false"): "ntt_pass_last", ("false", "false", "false"): "ntt_pass_single"}
log_n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
for k, v in summary.items():
    m = re.match(r"pm::ntt_pass4?_kernel<(\d+), (\d+), (\w+), (\w+), (\w+)>", k)
    if m and "hbm_bytes_per_launch_corrected" in v:
        traffic[f"{roles[(m.group(3), m.group(4), m.group(5))]}_2^{log_n}"] = int(v["hbm_bytes_per_launch_corrected"])
        if "SQ_INSTS_VALU" in v:
            traffic[f"{roles[(m.group(3), m.group(4), m.group(5))]}_2^{log_n}_valu_insts"] = int(v["SQ_INSTS_VALU"]["mean"])
for k, v in summary.items():
    if k.startswith("pm::msm_accumulate_l1") and "hbm_bytes_per_launch_corrected" in v:
        traffic[f"msm_accumulate_l1_2^{log_n}"] = int(v["hbm_bytes_per_launch_corrected"])
        if "SQ_INSTS_VALU" in v:
            traffic[f"msm_accumulate_l1_2^{log_n}_valu_insts"] = int(v["SQ_INSTS_VALU"]["mean"])
if traffic and prefix == "pmc_":
    json.dump(traffic, open(os.path.join(os.path.dirname(out), "ntt_traffic.json"), "w"), indent=1, sort_keys=True)
    print("traffic:", traffic)
for k, v in sorted(summary.items()):
    print(k, {c: (round(x["mean"]) if isinstance(x, dict) else round(x)) for c, x in v.items()})
