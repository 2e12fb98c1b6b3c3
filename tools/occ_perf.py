"""Diagnostic: 1 vs 2 workgroups per CU (1 vs 2 waves per SIMD) for a small-network fused kernel."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, '.')
from varnet_amd import engine
occ = sys.argv[1]
engine.LIB_PATH = os.path.join(os.path.dirname(engine.LIB_PATH), 'libvarnet_hip_occ%s.so' % occ)
from varnet_amd.engine import VNEngine
for L, H in ((2, 20), (3, 20)):
    d_in, dim, widths, q, n_k, nB = 3, 2, [H]*L, 64, 100000, 14000
    n = n_k*q
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    Input = torch.rand(n, d_in, device='cuda', generator=g)*2-1
    gcoef = torch.randn(n, dim, device='cuda', generator=g)
    eng = VNEngine(dim, d_in, widths, True, q)
    eng.init_params(0)
    rng = np.random.default_rng(0)
    eng.set_fe_table(rng.uniform(0,1,q), rng.standard_normal(q))
    eng.set_interior(0, Input, gcoef, None, n_k=n_k, detJ=1e-6)
    eng.set_bic(torch.rand(nB, d_in, device='cuda')*2-1, torch.randn(nB, device='cuda'), 9000, 2.0)
    eng.set_weights([1,1,1])
    loss = torch.zeros(1, device='cuda')
    for _ in range(3): eng.train_step(0, loss)
    torch.cuda.synchronize()
    eng.profile_begin()
    for _ in range(10): eng.train_step(0, loss)
    ms, nl, name = eng.profile_end()
    print('occ', occ, 'net %dx%d' % (L, H), 'kernel ms', ms, 'loss', loss.item())
    eng.close()
