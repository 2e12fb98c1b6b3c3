"""Step time at integNum = 216 (3-point Gauss, 2D+t): two-pass fused route (auto) vs generic kernels."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from varnet_amd.engine import VNEngine
d_in, dim, widths, q, n_k, nB = 3, 2, [50] * 5, 216, 30000, 14000
n = n_k * q
g = torch.Generator(device='cuda'); g.manual_seed(0)
Input = torch.rand(n, d_in, device='cuda', generator=g) * 2 - 1
gcoef = torch.randn(n, dim, device='cuda', generator=g)
rng = np.random.default_rng(0)
N1, dN1, W1 = rng.uniform(0, 1, q), rng.standard_normal(q), rng.uniform(0.3, 1, q)
bi = torch.rand(nB, d_in, device='cuda', generator=g) * 2 - 1
bl = torch.randn(nB, device='cuda', generator=g)
F_pt = 2 * (d_in * 50 + 4 * 2500 + 50)
for kernel, name in ((0, 'auto (two-pass fused)'), (1, 'generic')):
    e = VNEngine(dim, d_in, widths, True, q, integWflag=True, kernel=kernel)
    e.init_params(0); e.set_fe_table(N1, dN1, W1); e.set_interior(0, Input, gcoef, None, n_k=n_k, detJ=1e-6)
    e.set_bic(bi, bl, 9000, 2.0); e.set_weights([1, 1, 1])
    for _ in range(2): e.train_step(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    steps = 5
    for _ in range(steps): e.train_step(0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print('%-24s %.2f ms/step  %.3e points/s  %.1f TFLOP/s algorithmic (6 F_pt)' % (name, dt * 1e3, n / dt, 6 * F_pt * n / dt / 1e12))
    e.close()
