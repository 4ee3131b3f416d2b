#!/usr/bin/env python3
"""Condense a rocprofv3 run directory (gpurun_out/prof_rNN: kt/, pmc_fetch/, pmc_write/) into the small
files committed under profiles/:  <tag>_kernel_stats.csv, <tag>_k1_trace_summary.json, k1_pmc_traffic.json.
usage: summarize_profile.py gpurun_out/prof_r01 r01"""
import csv, glob, json, os, statistics, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
os.makedirs(root, exist_ok=True)
KEY = "OpProject"


def one(pattern):
    files = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    return files[-1]


# 1. --kernel-trace --stats summary (names shortened so the CSV stays readable)
rows = list(csv.DictReader(open(one("kt/*/*kernel_stats.csv"))))
with open(os.path.join(root, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])

# 2. per-dispatch durations of the dominant kernel
kt = list(csv.DictReader(open(one("kt/*/*kernel_trace.csv"))))
k1 = [r for r in kt if KEY in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in k1]
summary = {
    "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-secondary "
               "(50 warm-up launches, the 1000-launch graph once untimed, short-graph clock warm-up, the timed replay, one launch for the parity metric)",
    "kernel": k1[0]["Kernel_Name"][:120], "dispatches": len(d),
    "avg_us": statistics.mean(d), "median_us": statistics.median(d), "min_us": min(d), "max_us": max(d),
    "first_50_avg_us": statistics.mean(d[:50]), "last_100_avg_us": statistics.mean(d[-100:]),
    # the timed graph replay is the last block of 1000 dispatches before the single launch of the parity metric
    "timed_replay_avg_us": statistics.mean(d[-1001:-1]) if len(d) > 1001 else None,
    "timed_replay_median_us": statistics.median(d[-1001:-1]) if len(d) > 1001 else None,
    "vgpr": k1[0].get("VGPR_Count"), "sgpr": k1[0].get("SGPR_Count"), "lds_bytes": k1[0].get("LDS_Block_Size"),
    "grid": k1[0].get("Grid_Size"), "workgroup": k1[0].get("Workgroup_Size"),
    "algorithmic_bytes_per_launch": 72_000_000,
}
summary["achieved_GBps_at_avg"] = 72e6 / summary["avg_us"] * 1e-3
json.dump(summary, open(os.path.join(root, f"{tag}_k1_trace_summary.json"), "w"), indent=1)

# 3. HBM traffic from separate PMC passes; gfx950: FETCH_SIZE counts wide coalesced reads at half (MI355X_MICROARCH.md, HBM)
def counter(dirname, name):
    rows = list(csv.DictReader(open(one(f"{dirname}/*/*counter_collection.csv"))))
    v = [float(r["Counter_Value"]) for r in rows if KEY in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(v) / len(v), len(v)

fetch_kb, nf = counter("pmc_fetch", "FETCH_SIZE")
write_kb, nw = counter("pmc_write", "WRITE_SIZE")
traffic = {
    "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB_raw": write_kb, "dispatches_fetch": nf, "dispatches_write": nw,
    "correction": "gfx950: FETCH_SIZE reports half of a wide coalesced streaming read -> doubled; WRITE_SIZE exact",
    "hbm_read_bytes_per_launch": 2 * fetch_kb * 1024, "hbm_write_bytes_per_launch": write_kb * 1024,
    "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024, "algorithmic_bytes_per_launch": 72_000_000,
    "source": f"profiles/{tag}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 20 --warmup 5",
}
json.dump(traffic, open(os.path.join(root, "k1_pmc_traffic.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(root, f"{tag}_k1_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)); print(json.dumps(traffic, indent=1))

# 4. every kernel of the library under --kernel-trace --stats (tools/bench_all.py), if that pass was made
try:
    rows = list(csv.DictReader(open(one("kt_all/*/*kernel_stats.csv"))))
    with open(os.path.join(root, f"{tag}_all_kernels_stats.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            if any(k in r["Name"] for k in ("so3::", "k_kabsch", "k_add_l1", "k_rotate_clouds", "k_pc_normalize", "k_stats", "k_project", "k_frob", "k_angle", "k_geodesic", "k_op_rows")):
                w.writerow([" ".join(r["Name"].split())[:140], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    print("wrote", f"{tag}_all_kernels_stats.csv")
except (IndexError, OSError) as exc:
    print("no kt_all pass:", exc)
