import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
from poseestimation_amd import rotation_representation as rr
from oracle import c_oracle
rng=np.random.default_rng(0)
n=400000
from oracle import so3_oracle as so
U=so.symmetric_orthogonalization_np(rng.standard_normal((n,9))); V=so.symmetric_orthogonalization_np(rng.standard_normal((n,9)))
k2=rng.uniform(0,7,n); k3=k2+rng.uniform(0,3,n)
sgn=np.where(rng.random(n)<0.5,-1.0,1.0)
S=np.stack([np.ones(n),10**-k2,sgn*10**-k3],1)
M=(U*S[:,None,:])@V.transpose(0,2,1)
x=torch.tensor(M.reshape(n,9),dtype=torch.float32,device='cuda')
r=rr.symmetric_orthogonalization(x).cpu().numpy().astype(np.float64)
orth=np.linalg.norm(np.einsum('bji,bjk->bik',r,r)-np.eye(3),axis=(1,2))
ref=c_oracle.project(x.cpu().numpy())
err=np.abs(r-ref).reshape(n,-1).max(1)
print('ill-conditioned batch: orth max %.3e  p99.9 %.3e ; nan %d'%(orth.max(),np.quantile(orth,.999),np.isnan(r).sum()))
for lo,hi in ((0,2),(2,4),(4,6),(6,7)):
    m=(k2>=lo)&(k2<hi)
    print('  s2 in 1e-[%d,%d): orth max %.2e  err median %.2e p99 %.2e'%(lo,hi,orth[m].max(),np.median(err[m]),np.quantile(err[m],.99)))
