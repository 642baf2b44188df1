"""profiles/pmc_traffic.json from a pmc_hbm_traffic.csv of scripts/collect_profiles.sh: the HBM bytes per launch of the
kernel bench.py's `roofline` object is about (the H contraction), FETCH_SIZE already doubled by pmc_summary.py as
MI355X_MICROARCH.md prescribes for gfx950.  usage: make_pmc_traffic.py profiles/r03x_pmc_hbm_traffic.csv"""
import csv, json, os, sys
src = sys.argv[1]
rows = []
for ln in open(src).read().splitlines()[1:]:
    parts = ln.rsplit(",", 4)          # (template arguments put commas into the kernel name)
    if len(parts) == 5 and "hk_panel" in parts[0]:
        rows.append({"kernel": parts[0], "launches": parts[1], "fetch_MB_per_launch(x2 corrected)": parts[2],
                     "write_MB_per_launch": parts[3], "total_MB_per_launch": parts[4]})
assert rows, "no hk_panel kernel in " + src
r = max(rows, key=lambda q: float(q["total_MB_per_launch"]))
out = {"hk_panel_kernel": {
    "kernel_name": r["kernel"], "launches": int(r["launches"]),
    "hbm_bytes_per_launch": int(float(r["total_MB_per_launch"]) * 1e6),
    "fetch_bytes_per_launch_x2_corrected": int(float(r["fetch_MB_per_launch(x2 corrected)"]) * 1e6),
    "write_bytes_per_launch": int(float(r["write_MB_per_launch"]) * 1e6),
    "source": f"{src} (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes with --kernel-trace only, "
              "bench.py --steps 1 --warmup 0; FETCH_SIZE doubled per MI355X_MICROARCH.md)"}}
json.dump(out, open(os.path.join(os.path.dirname(src), "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
