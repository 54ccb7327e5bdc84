#!/usr/bin/env python3
"""Copy what scripts/make_profiles.sh left under gpurun_out/profiles/ into profiles/<round>/ and rebuild traffic.json
(the measured HBM bytes per launch of each workload's dominant K kernel, which bench.py reports as roofline.traffic).
Usage: python scripts/collect_profiles.py r03"""
import json, os, shutil, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = "gpurun_out/profiles"
dst = os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)
# tag of make_profiles.sh -> (workload, variant, [kernel name patterns as rocprof prints them], label bench.py uses, nel)
DOM = {
    "elastic3d": ("elastic3d", "default", ["k_assemble_s3_unit"], "k_assemble_s3_unit", 128),
    "poisson3d": ("poisson3d", "default", ["k_assemble_p3_rect"], "k_assemble_p3_rect", 256),
    "elastic2d": ("elastic2d", "default", ["k_assemble_e2_rect"], "k_assemble_e2_rect", 2048),
    "elastic3d_manufactured": ("elastic3d", "manufactured", ["k_assemble_s3_unit"], "k_assemble_s3_unit", 128),
    "elastic3d_general": ("elastic3d", "general", ["k_assemble_s3_hex2"], "k_assemble_s3_hex2", 128),
    "poisson3d_general": ("poisson3d", "general", ["k_assemble_p3_hex4"], "k_assemble_p3_hex4", 256),
    "elastic2d_general": ("elastic2d", "general", ["k_assemble_ke_e2"], "k_assemble_ke_e2", 2048),
    "enriched3d": ("enriched3d", "default", ["k_enr_box<2>"], "k_enr_box<S=2>", 128),
    "enriched3d_general": ("enriched3d", "general", ["k_enr_node<2; true>"], "k_enr_node<S=2,one material per element>", 128),
    # the two-kernel form: all of K = element matrices + segmented reduction, bytes per launch PAIR times the launch
    # pairs of a step
    "enriched3d_valu": ("enriched3d", "valu", ["k_enr_elem<2; false>", "k_enr_gather<6>"], "k_enr_elem + k_enr_gather", 128),
    "enriched3d_mfma": ("enriched3d", "mfma", ["k_enr_elem<2; true>", "k_enr_gather<6>"], "k_enr_elem + k_enr_gather", 128),
}
kernels = []
for tag, (wl, variant, pats, label, nel) in DOM.items():
    if not os.path.exists(f"{src}/{tag}_bench.json"):
        continue
    shutil.copy(f"{src}/{tag}_bench.json", f"{dst}/bench_{tag}.json")
    shutil.copy(f"{src}/{tag}_kernel_stats.csv", f"{dst}/bench_{tag}_kernel_stats.csv")
    rows = open(f"{src}/{tag}_pmc_WRITE_SIZE.txt").read() + open(f"{src}/{tag}_pmc_FETCH_SIZE.txt").read()
    open(f"{dst}/bench_{tag}_pmc_hbm.csv", "w").write(rows)
    wr = rd = 0.0
    per_kernel = {}
    launches = 1
    for pat in pats:
        val, cnt = {}, {}
        for line in rows.splitlines():
            c, rest = line.split(",", 1)
            name, n, mean = rest.rsplit(",", 2)
            if pat in name:
                val[c] = float(mean)
                cnt[c] = int(n)
        per_kernel[pat] = {"WRITE_SIZE_KiB": val["WRITE_SIZE"], "FETCH_SIZE_KiB": val["FETCH_SIZE"]}
        wr += 1024.0 * val["WRITE_SIZE"]
        rd += 2.0 * 1024.0 * val["FETCH_SIZE"]
        if len(pats) > 1:
            launches = max(launches, cnt["WRITE_SIZE"] // 7)  # the PMC passes run 2 warm-up + 5 timed steps
    kernels.append({"workload": wl, "variant": variant, "nel": nel, "kernel": label, "counters": per_kernel,
                    "launches_per_step": launches, "hbm_write_bytes": wr * launches,
                    "hbm_read_bytes_corrected_x2": rd * launches, "hbm_bytes_per_launch": (wr + rd) * launches})
out = {
    "source": f"profiles/{rnd}/bench_*_pmc_hbm.csv: separate rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE passes of "
              "`python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline [--workload W ...]`, mean over the dispatches of "
              "the dominant kernel (scripts/make_profiles.sh, scripts/collect_profiles.py).  For the enriched workload "
              "the roofline is stated for all of K, so the entry is the sum over the step's launches of both kernels.",
    "correction": "bytes = 1024*WRITE_SIZE + 2*1024*FETCH_SIZE (gfx950 FETCH_SIZE counts 128-B requests as 64 B; the "
                  "factor is calibrated for 16-B-per-lane streaming reads and is an upper estimate for 8-B loads)",
    "kernels": kernels,
}
json.dump(out, open(f"{dst}/traffic.json", "w"), indent=1)
# the committed bench lines carry the traffic of THESE counter passes (the stats run printed its line before
# traffic.json of this round existed, i.e. with the previous round's figure): same command, same build, same call
by = {(k["workload"], k["variant"]): k["hbm_bytes_per_launch"] for k in kernels}
for tag, (wl, variant, pats, label, nel) in DOM.items():
    f = f"{dst}/bench_{tag}.json"
    if not os.path.exists(f):
        continue
    lines = open(f).read().strip().splitlines()
    d = json.loads(lines[-1])
    d["roofline"]["traffic"] = by[(wl, variant)]
    gg = d.get("general_geometry")
    if gg and (wl, "general") in by:
        gg["roofline"]["traffic"] = by[(wl, "general")]
    open(f, "w").write(json.dumps(d) + "\n")
for k in kernels:
    print(k["workload"], k["variant"], k["kernel"], "%.4g B per step-launch" % k["hbm_bytes_per_launch"])
