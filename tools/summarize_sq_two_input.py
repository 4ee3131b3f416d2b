#!/usr/bin/env python3
"""Per-dispatch SQ counters of K2 / K3 / K1+K4 from tools/profile_round.sh's passes (sqa_<k>, sqb_<k> over tools/kernel_loop.py).
usage: summarize_sq_two_input.py gpurun_out/prof_r05 r05   ->  docs/history/profiles/r05_two_input_pmc_sq.json"""
import csv, glob, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
names = {"k2": "OpProjectBwd", "k3": "OpFrobHead", "k14": "OpProjectAngle"}
rounds = (1_000_000 + 127) // 128
out = {"source": "rocprofv3 --pmc <set> (two separate passes per kernel) -- python3 tools/kernel_loop.py k2|k3|k14 12 ; per-dispatch averages, "
                 "1M Gaussian rows, %d rounds of 128 rows" % rounds, "kernels": {}}
for k, key in names.items():
    vals = {}
    for d in ("sqa_" + k, "sqb_" + k):
        for f in glob.glob(os.path.join(src, d, "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if key in r["Kernel_Name"]:
                    vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                    kernel = " ".join(r["Kernel_Name"].split())[:120]
    if not vals:
        continue
    c = {n: sum(v) / len(v) for n, v in vals.items()}
    out["kernels"][k] = {"kernel": kernel, "counters": c, "derived": {
        "valu_instructions_per_round": c["SQ_INSTS_VALU"] / rounds, "salu_per_round": c["SQ_INSTS_SALU"] / rounds,
        "lds_instructions_per_round": c["SQ_INSTS_LDS"] / rounds, "transcendental_per_round": c["SQ_INSTS_VALU_TRANS_F32"] / rounds,
        "valu_active_fraction_of_wave_cycles": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
        "waiting_fraction_of_wave_cycles_WAIT_INST_ANY": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]}}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", f"{tag}_two_input_pmc_sq.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v["derived"] for k, v in out["kernels"].items()}, indent=1))
