"""Kernel time of AUTO vs the generic kernels for a given width/depth on config-3 sized inputs:
   python tools/width_perf.py H L [n_k]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from varnet_amd.engine import VNEngine
H, L = int(sys.argv[1]), int(sys.argv[2])
n_k = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
d_in, dim, widths, q, nB = 3, 2, [H] * L, 64, 14000
n = n_k * q
g = torch.Generator(device='cuda'); g.manual_seed(0)
Input = torch.rand(n, d_in, device='cuda', generator=g) * 2 - 1
gcoef = torch.randn(n, dim, device='cuda', generator=g)
bi = torch.rand(nB, d_in, device='cuda', generator=g) * 2 - 1
bl = torch.randn(nB, device='cuda', generator=g)
rng = np.random.default_rng(0)
N1, dN1 = rng.uniform(0, 1, q), rng.standard_normal(q)
Fpt = 2 * (d_in * H + (L - 1) * H * H + H)
for kernel in (0, 1):
    e = VNEngine(dim, d_in, widths, True, q, kernel=kernel)
    e.init_params(0); e.set_fe_table(N1, dN1); e.set_interior(0, Input, gcoef, None, n_k=n_k, detJ=1e-6)
    e.set_bic(bi, bl, 9000, 2.0); e.set_weights([1, 1, 1])
    for _ in range(2): e.train_step(0)
    e.profile_begin()
    for _ in range(5): e.train_step(0)
    ms, nl, kn = e.profile_end()
    print('H=%d L=%d kernel=%s path=%s: %s %.3f ms  -> %.3f of fp32 MFMA peak (6 F_pt)' %
          (H, L, 'auto' if kernel == 0 else 'generic', e.kernel_path(), kn, ms, (6 * Fpt * n + 3 * Fpt * nB) / (ms * 1e-3) / 157.3e12))
    e.close()
