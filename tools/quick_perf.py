import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from varnet_amd.engine import VNEngine
d_in, dim, widths, integNum, n_k, nB = 3, 2, [50]*5, 64, 100000, 14000
n = n_k*integNum
g = torch.Generator(device='cuda'); g.manual_seed(0)
Input = torch.rand(n, d_in, device='cuda', generator=g)*2-1
gcoef = torch.randn(n, dim, device='cuda', generator=g)
biInput = torch.rand(nB, d_in, device='cuda', generator=g)*2-1
biLabel = torch.randn(nB, device='cuda', generator=g)
eng = VNEngine(dim, d_in, widths, True, integNum, kernel=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
eng.init_params(0)
rng = np.random.default_rng(0)
eng.set_fe_table(rng.uniform(0,1,integNum), rng.standard_normal(integNum))
eng.set_interior(0, Input, gcoef, None, n_k=n_k, detJ=1e-6)
eng.set_bic(biInput, biLabel, 9000, 2.0)
eng.set_weights([1,1,1])
loss = torch.zeros(1, device='cuda')
for _ in range(3): eng.train_step(0, loss)
torch.cuda.synchronize()
K = 10
eng.profile_begin()
t = time.time()
for _ in range(K): eng.train_step(0, loss)
torch.cuda.synchronize()
dt = (time.time()-t)/K
ms, nl, name = eng.profile_end()
Fpt = 2*(3*50+4*2500+50)
print('ms/step', dt*1e3, 'pts/s', n/dt, 'frac of 157TF (6F)', 6*Fpt*n/dt/157.3e12, name, ms, nl, 'loss', loss.item())
