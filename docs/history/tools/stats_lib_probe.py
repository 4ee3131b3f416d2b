import ctypes, sys, torch
lib = ctypes.CDLL(sys.argv[1])
P = ctypes.c_void_p
lib.so3_angle_stats.restype = ctypes.c_int
lib.so3_angle_stats.argtypes = [P, P, ctypes.c_int32, P, P, ctypes.c_int64, P]
lib.so3_angle_stats_workspace_bytes.restype = ctypes.c_size_t
n, ncls = 1_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 10
deg = torch.rand(n, device="cuda", dtype=torch.float64) * 180
cls = torch.randint(0, ncls, (n,), device="cuda", dtype=torch.int32)
stats = torch.empty(ncls, 8, dtype=torch.float64, device="cuda")
work = torch.empty(lib.so3_angle_stats_workspace_bytes(), dtype=torch.uint8, device="cuda")
st = P(torch.cuda.current_stream().cuda_stream)
for _ in range(20):
    lib.so3_angle_stats(P(deg.data_ptr()), P(cls.data_ptr()), ncls, P(stats.data_ptr()), P(work.data_ptr()), n, st)
torch.cuda.synchronize()
