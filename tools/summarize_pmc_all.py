#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/bench_all.py.
usage: summarize_pmc_all.py gpurun_out/pmc_all r01   ->  docs/history/profiles/r01_all_kernels_pmc_traffic.json
gfx950: FETCH_SIZE reports half of a wide coalesced streaming read (MI355X_MICROARCH.md) -> doubled."""
import collections, csv, glob, json, os, re, sys

src, tag = sys.argv[1], sys.argv[2]


def load(sub, name):
    f = sorted(glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            acc[(r["Kernel_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return acc


fetch, write = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
out = []
for key in fetch:
    name, grid = key
    if not re.search(r"so3::|k_kabsch|k_add_l1|k_rotate_clouds|k_pc_normalize|k_stats|k_op_rows|k_project|k_frob|k_angle|k_geodesic", name):
        continue
    f = fetch[key]; w = write.get(key, [0.0])
    out.append({"kernel": re.sub(r"\s+", " ", name)[:150], "grid": grid, "dispatches": len(f),
                "hbm_read_MB": round(2 * sum(f) / len(f) * 1024 / 1e6, 3), "hbm_write_MB": round(sum(w) / len(w) * 1024 / 1e6, 3)})
out.sort(key=lambda d: -(d["hbm_read_MB"] + d["hbm_write_MB"]))
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", f"{tag}_all_kernels_pmc_traffic.json")
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/bench_all.py with SO3_BENCH_QUICK=1",
           "note": "FETCH_SIZE doubled (gfx950 caveat); per-dispatch averages; 1M rows per call unless the kernel works on clouds (65536 x 1024 points)",
           "kernels": out}, open(dst, "w"), indent=1)
for d in out:
    print("%8.1f MB read %8.1f MB written  x%-3d %s" % (d["hbm_read_MB"], d["hbm_write_MB"], d["dispatches"], d["kernel"][:100]))
