import sys, time, torch
sys.path.insert(0,'/root/repo')
from poseestimation_amd import rotation_representation as rr
dev='cuda:0'
class Triv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save = None
        return x.new_empty(())
    @staticmethod
    def backward(ctx, g):
        return None
class Triv2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, buf):
        ctx.buf = buf
        return x.new_empty(())
    @staticmethod
    def backward(ctx, g):
        return ctx.buf, None
x=torch.randn(512,9,device=dev,dtype=torch.bfloat16,requires_grad=True)
buf=torch.zeros_like(x)
t=torch.eye(3,device=dev).repeat(512,1,1)
def bench(fn,n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
def f_triv():
    x.grad=None; Triv.apply(x).backward()
def f_triv2():
    x.grad=None; Triv2.apply(x,buf).backward()
def f_ours():
    x.grad=None; loss,r=rr.frobenius_head(x,t); loss.backward()
def f_ours_nor():
    x.grad=None; loss=rr.frobenius_head(x,t,return_rotation=False); loss.backward()
def f_fwd():
    rr.frobenius_head(x,t)
def f_native():
    x.grad=None; (x.float().sum()).backward()
print("trivial Function (no grad returned) + backward: %.1f us"%bench(f_triv))
print("trivial Function returning a stored buffer + backward: %.1f us"%bench(f_triv2))
print("x.float().sum().backward() (two native nodes): %.1f us"%bench(f_native))
print("frobenius_head + backward: %.1f us"%bench(f_ours))
print("frobenius_head(return_rotation=False) + backward: %.1f us"%bench(f_ours_nor))
print("frobenius_head forward only (requires_grad): %.1f us"%bench(f_fwd))
# does a live hipGraph / a side stream / big allocations in the process change the cost?
big = [torch.empty(36_000_000 // 4 * 8, device=dev) for _ in range(2)]
print("after 576 MB of allocations: %.1f us" % bench(f_ours))
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
y = torch.zeros(1024, device=dev)
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        y.add_(1.0)
torch.cuda.synchronize()
print("after capturing a graph on a side stream: %.1f us" % bench(f_ours))
with torch.cuda.stream(side):
    for _ in range(50): g.replay()
torch.cuda.synchronize()
print("after replaying it: %.1f us" % bench(f_ours))
del g
torch.cuda.synchronize()
print("after deleting the graph: %.1f us" % bench(f_ours))
