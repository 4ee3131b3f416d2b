#!/usr/bin/env python3
"""Where a call of so3_angle_stats spends its time: wall-clock stamps (s_memrealtime) of workgroups 0 (a finishing one) and 100 at the
phase boundaries of both kernels, for 10 classes / one class and uniform angles / K4's angles between random rotations.
Needs a build with the stamps compiled in:   tools/build_variant.sh stamp -DSO3_STATS_STAMP
usage: stats_anatomy.py build/variants/libso3proj_stamp.so [replicas (SO3_STAT_REPLICAS of that build: 4)]"""
import ctypes, sys, os, torch
sys.path.insert(0, os.getcwd())
from poseestimation_amd import rotation_representation as rr
lib = ctypes.CDLL(sys.argv[1])
P = ctypes.c_void_p
lib.so3_angle_stats.argtypes = [P, P, ctypes.c_int, P, P, ctypes.c_int64, P]
lib.so3_angle_stats_workspace_bytes.restype = ctypes.c_size_t
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 1_000_000
st = P(torch.cuda.current_stream().cuda_stream)
wb = lib.so3_angle_stats_workspace_bytes()
OFF = R * 2048 + 16 + 256 + R * 64 * 516 * 4 + (1 << 20)          # offsetof(StatWork, cand): the stamps sit in the buffer's last 64 entries
names = ["start", "selected", "rows staged", "candidates out", "tickets in", "class done", "", "", "fin: sync", "fin: cached", "fin: first vote", "fin: selected"]
for ncls, kind in ((10, "uniform"), (10, "haar"), (1, "uniform"), (1, "haar")):
    if kind == "haar":
        a, b = (rr.symmetric_orthogonalization(torch.randn(n, 9, device="cuda")) for _ in range(2))
        deg = rr.angle_error(a, b).to(torch.float64)
    else:
        deg = torch.rand(n, device="cuda", dtype=torch.float64) * 180
    cls = torch.randint(0, ncls, (n,), device="cuda", dtype=torch.int32)
    stats = torch.empty(ncls, 8, dtype=torch.float64, device="cuda")
    work = torch.zeros(wb, dtype=torch.uint8, device="cuda")
    for _ in range(5):
        rc = lib.so3_angle_stats(P(deg.data_ptr()), P(cls.data_ptr()), ncls, P(stats.data_ptr()), P(work.data_ptr()), n, st)
        assert rc == 0
    torch.cuda.synchronize()
    t = work[OFF + ((1 << 20) - 64) * 8: OFF + (1 << 20) * 8].view(torch.int64).cpu().tolist()
    for blk, base in ((0, 0), (100, 32)):
        s = t[base:base + 12]
        line = ["%s %.2f" % (names[i], (s[i] - s[0]) * 0.01) for i in (1, 2, 3, 4, 8, 9, 10, 11, 5) if s[i]]
        print("%2d classes %-8s wg %3d: " % (ncls, kind, blk) + " | ".join(line))
        ww = t[base + 16:base + 21]
        print("      window wg %3d: lds zeroed %.2f | rows %.2f | flushed (issued) %.2f | (performed) %.2f | -> collect's start %.2f" % ((blk,) + tuple((ww[i] - ww[0]) * 0.01 for i in (1, 2, 3, 4)) + ((s[0] - ww[0]) * 0.01,)))
