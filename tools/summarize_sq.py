#!/usr/bin/env python3
"""Per-dispatch averages of K1's SQ counters from separate rocprofv3 --pmc passes over tools/k1_loop.py.
usage: summarize_sq.py gpurun_out/prof_r02 r02   (reads <dir>/sq1, sq2, sq3; writes profiles/<tag>_k1_pmc_sq.json)"""
import csv, glob, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
vals = {}
for d in ("sq1", "sq2", "sq3"):
    for f in glob.glob(os.path.join(src, d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "OpProject" in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in vals.items()}
rounds = (1_000_000 + 127) // 128
out = {"source": "rocprofv3 --pmc <set> (three separate passes) -- python3 tools/k1_loop.py 20 ; per-dispatch averages of K1 "
                 "(1M Gaussian rows, 3072 waves, %d rounds of 128 rows)" % rounds,
       "counters": c,
       "derived": {"valu_instructions_per_round": c["SQ_INSTS_VALU"] / rounds,
                   "fma_per_round": c["SQ_INSTS_VALU_FMA_F32"] / rounds, "mul_per_round": c["SQ_INSTS_VALU_MUL_F32"] / rounds,
                   "add_per_round": c["SQ_INSTS_VALU_ADD_F32"] / rounds, "transcendental_per_round": c["SQ_INSTS_VALU_TRANS_F32"] / rounds,
                   "lds_instructions_per_round": c["SQ_INSTS_LDS"] / rounds, "salu_per_round": c["SQ_INSTS_SALU"] / rounds,
                   "vmem_read_per_round": c["SQ_INSTS_VMEM_RD"] / rounds, "vmem_write_per_round": c["SQ_INSTS_VMEM_WR"] / rounds,
                   "valu_active_fraction_of_wave_cycles": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
                   "waiting_fraction_of_wave_cycles_WAIT_INST_ANY": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                   "parked_fraction_of_wave_cycles_WAIT_ANY": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]}}
json.dump(out, open(os.path.join(root, f"{tag}_k1_pmc_sq.json"), "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
