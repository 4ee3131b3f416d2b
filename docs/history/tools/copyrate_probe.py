import torch
x=[torch.randn(65536*1024*3, device='cuda') for _ in range(2)]
y=[torch.empty_like(x[0]) for _ in range(2)]
for i in range(3): y[i%2].copy_(x[i%2])
torch.cuda.synchronize()
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for i in range(10): y[i%2].copy_(x[i%2])
b.record(); torch.cuda.synchronize()
us=a.elapsed_time(b)*1e3/10
print("torch copy 805MB->805MB: %.1f us  %.0f GB/s"%(us, 2*x[0].numel()*4/us*1e-3))
a.record()
for i in range(10): torch.add(x[i%2], 1.0, out=y[i%2])
b.record(); torch.cuda.synchronize()
us=a.elapsed_time(b)*1e3/10
print("torch add  805MB->805MB: %.1f us  %.0f GB/s"%(us, 2*x[0].numel()*4/us*1e-3))
