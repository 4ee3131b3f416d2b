import sys, time, torch
sys.path.insert(0,'/root/repo')
from poseestimation_amd import rotation_representation as rr
import cProfile, pstats
dev='cuda:0'
b=512
x=torch.randn(b,9,device=dev).bfloat16().requires_grad_(True)
rt=rr.symmetric_orthogonalization(torch.randn(b,9,device=dev))
def it():
    loss,_=rr.frobenius_head(x,rt); loss.backward(); x.grad=None
for _ in range(20): it()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(200): it()
torch.cuda.synchronize()
print('fwd+bwd via mirror: %.1f us/iter'%((time.perf_counter()-t0)/200*1e6))
pr=cProfile.Profile(); pr.enable()
for _ in range(200): it()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
xf=torch.randn(b,9,device=dev)
for _ in range(20): rr.symmetric_orthogonalization(xf)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(500): rr.symmetric_orthogonalization(xf)
torch.cuda.synchronize(); print('forward only via mirror: %.1f us/call'%((time.perf_counter()-t0)/500*1e6))
