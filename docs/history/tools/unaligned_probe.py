import sys, torch, ctypes
sys.path.insert(0, ".")
from poseestimation_amd import rotation_representation as rr
n = 1_000_000
x = torch.randn(n + 8, 9, device="cuda")
def t(fn, reps=300):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for off in (0, 1, 2, 3, 4):
    v = x[off:off + n]
    r = rr.symmetric_orthogonalization(v)
    ref = rr.symmetric_orthogonalization(v.clone())
    print("row offset %d (pointer %% 16 = %d): max |diff| vs aligned copy %.1e, %.2f us per call" % (off, v.data_ptr() % 16, (r - ref).abs().max().item(), t(lambda: rr.symmetric_orthogonalization(v))))
