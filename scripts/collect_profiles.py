#!/usr/bin/env python3
"""Copy what scripts/make_profiles.sh left under gpurun_out/profiles/ into profiles/<round>/ and rebuild traffic.json
(the measured HBM bytes per launch of each workload's dominant K kernel, which bench.py reports as roofline.traffic).
Usage: python scripts/collect_profiles.py r01"""
import json, os, shutil, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = "gpurun_out/profiles"
dst = os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)
# workload tag -> (template name rocprof prints, label bench.py uses, nel)
DOM = {
    "elastic3d": ("k_assemble_rowsweep<2>", "k_assemble_rowsweep<ELASTIC3D>", 128),
    "poisson3d": ("k_assemble_p3_rect", "k_assemble_p3_rect", 256),
    "elastic2d": ("k_assemble_e2_rect", "k_assemble_e2_rect", 2048),
}
kernels = []
for tag, (pat, label, nel) in DOM.items():
    if not os.path.exists(f"{src}/{tag}_bench.json"):
        continue
    shutil.copy(f"{src}/{tag}_bench.json", f"{dst}/bench_{tag}.json")
    shutil.copy(f"{src}/{tag}_kernel_stats.csv", f"{dst}/bench_{tag}_kernel_stats.csv")
    rows = open(f"{src}/{tag}_pmc_WRITE_SIZE.txt").read() + open(f"{src}/{tag}_pmc_FETCH_SIZE.txt").read()
    open(f"{dst}/bench_{tag}_pmc_hbm.csv", "w").write(rows)
    val = {}
    for line in rows.splitlines():
        c, rest = line.split(",", 1)
        name, n, mean = rest.rsplit(",", 2)
        if pat in name:
            val[c] = float(mean)
    wr = 1024.0 * val["WRITE_SIZE"]
    rd = 2.0 * 1024.0 * val["FETCH_SIZE"]
    kernels.append({"workload": tag, "nel": nel, "kernel": label, "WRITE_SIZE_KiB": val["WRITE_SIZE"],
                    "FETCH_SIZE_KiB": val["FETCH_SIZE"], "hbm_write_bytes": wr, "hbm_read_bytes_corrected_x2": rd,
                    "hbm_bytes_per_launch": wr + rd})
out = {
    "source": f"profiles/{rnd}/bench_*_pmc_hbm.csv: separate rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE passes of "
              "`python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline [--workload W]`, mean over the dispatches of "
              "the dominant kernel (scripts/make_profiles.sh, scripts/collect_profiles.py)",
    "correction": "bytes = 1024*WRITE_SIZE + 2*1024*FETCH_SIZE (gfx950 FETCH_SIZE counts 128-B requests as 64 B; the "
                  "factor is calibrated for 16-B-per-lane streaming reads and is an upper estimate for 8-B loads)",
    "kernels": kernels,
}
json.dump(out, open(f"{dst}/traffic.json", "w"), indent=1)
for k in kernels:
    print(k["workload"], k["kernel"], "%.4g B/launch" % k["hbm_bytes_per_launch"])
