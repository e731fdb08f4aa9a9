"""profiles/rNN_pmc_valu.json: VALU issue counters of the dominant kernel of each bench configuration, from the PMC
summaries of scripts/gpu_step_pmc.sh (`name=summary.txt:kernel-substring:units` arguments)."""
import json
import re
import sys

out = sys.argv[1]
res = {}
for spec in sys.argv[2:]:
    cfg, path, sub, units = spec.split(":")
    name, cur = None, {}
    best = None
    for line in open(path):
        m = re.match(r"^(tq_.*?)\s+grid=", line)
        if m:
            name = m.group(1)
            cur = {}
            if sub in name:
                best = (name, cur)
            continue
        m = re.match(r"^\s+(\w+)\s+(\d+)", line)
        if m and name:
            cur[m.group(1)] = int(m.group(2))
    if best:
        name, c = best
        res[cfg] = {"kernel": name, "units": int(units), "SQ_INSTS_VALU": c.get("SQ_INSTS_VALU"),
                    "SQ_INSTS_VALU_TRANS_F32": c.get("SQ_INSTS_VALU_TRANS_F32"), "SQ_ACTIVE_INST_VALU": c.get("SQ_ACTIVE_INST_VALU"),
                    "GRBM_GUI_ACTIVE": c.get("GRBM_GUI_ACTIVE"),
                    "valu_busy": c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (c["GRBM_GUI_ACTIVE"] / 8)}
json.dump({"source": "rocprofv3 --kernel-trace --pmc ... (separate passes, scripts/gpu_step_pmc.sh) over bench.py --quick on MI355X; "
                     "summaries in profiles/ next to this file",
           "valu_busy": "SQ_ACTIVE_INST_VALU (quad-cycles, summed over the chip) x 4 / 1024 SIMDs, over GRBM_GUI_ACTIVE (summed over "
                        "8 XCDs) / 8: the fraction of the launch during which a SIMD is issuing a VALU instruction",
           "configs": res}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
