"""profiles/rNN_pmc_step_traffic.json from the summaries of scripts/gpu_step_pmc.sh (PMC passes over `bench.py --quick`):
per config, the HBM bytes per launch of every kernel a step launches, and their sum.
usage: make_step_traffic.py out.json cfg=summary.txt:units[:kernel-substring,kernel-substring...] ..."""
import json
import re
import sys

out = sys.argv[1]
configs = {}
for arg in sys.argv[2:]:
    cfg, rest = arg.split("=", 1)
    parts = rest.split(":")
    path, units = parts[0], int(parts[1])
    want = parts[2].split(",") if len(parts) > 2 else ["tq_sample_locals_tail_kernel", "tq_pixel_unit_kernel"]
    kernels, name, grid = {}, None, None
    for line in open(path):
        m = re.match(r"^(tq_.*?)\s+grid=(\d+)", line)
        if m:
            name, grid = m.group(1), int(m.group(2))
            continue
        m = re.match(r"^\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+\(n=(\d+)\)", line)
        if m and name:
            key = next((w for w in want if w in name), None)
            if key is None:
                continue
            k = kernels.setdefault(key, {"kernel": name, "grid": grid})
            if k["grid"] != grid:  # the same kernel at another size (e.g. the autotuner's launches): keep the larger grid
                if grid < k["grid"]:
                    continue
                k.clear()
                k.update(kernel=name, grid=grid)
            k[m.group(1)] = int(m.group(2))
            k["launches_sampled"] = int(m.group(3))
    for k in kernels.values():
        if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
            k["fetch_bytes"] = k["FETCH_SIZE"] * 1024 * 2
            k["write_bytes"] = k["WRITE_SIZE"] * 1024
            k["traffic_bytes"] = k["fetch_bytes"] + k["write_bytes"]
    configs[cfg] = {"units": units, "kernels": kernels,
                    "step_traffic_bytes": sum(k.get("traffic_bytes", 0) for k in kernels.values()), "summary": path}
json.dump({
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/gpu_step_pmc.sh) over "
              "`bench.py --quick` on MI355X: median per kernel over the launches of the run; summaries in profiles/ next to this file",
    "correction": "FETCH_SIZE is reported in KiB and counts 128-B requests as 64 B on gfx950 (MI355X_MICROARCH.md, HBM section): "
                  "bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE * 1024 is exact",
    "configs": configs}, open(out, "w"), indent=1)
print(json.dumps({c: {"step": v["step_traffic_bytes"], **{k: x.get("traffic_bytes") for k, x in v["kernels"].items()}} for c, v in configs.items()}, indent=1))
