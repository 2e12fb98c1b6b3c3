"""Time of vn_eval_loss (splitLoss) and vn_forward on config-3 sized inputs: AUTO (fused forward-only mode) vs generic."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from varnet_amd.engine import VNEngine
d_in, dim, widths, q, n_k, nB = 3, 2, [50] * 5, 64, 100000, 14000
n = n_k * q
g = torch.Generator(device='cuda'); g.manual_seed(0)
X = torch.rand(n, d_in, device='cuda', generator=g) * 2 - 1
G = torch.randn(n, dim, device='cuda', generator=g)
bi = torch.rand(nB, d_in, device='cuda', generator=g) * 2 - 1
bl = torch.randn(nB, device='cuda', generator=g)
rng = np.random.default_rng(0)
N1, dN1 = rng.uniform(0, 1, q), rng.standard_normal(q)
res = {}
for kernel in (0, 1):
    e = VNEngine(dim, d_in, widths, True, q, kernel=kernel)
    e.init_params(0); e.set_fe_table(N1, dN1); e.set_interior(0, X, G, None, n_k=n_k, detJ=1e-6)
    e.set_bic(bi, bl, 9000, 2.0); e.set_weights([1, 1, 1])
    out, lv = e.eval_loss(0, lossVec=True); u = e.forward(X)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): out, lv = e.eval_loss(0, lossVec=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(5): u = e.forward(X)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    res[kernel] = (out, lv.double().sum().item(), u.double().sum().item())
    print('%-8s eval_loss %.2f ms   forward(6.4M rows) %.2f ms   loss %.6e' % ('auto' if kernel == 0 else 'generic', (t1 - t0) / 5 * 1e3, (t2 - t1) / 5 * 1e3, out[0]))
    e.close()
a, b = res[0], res[1]
print('auto vs generic: loss rel %.1e  lossVec-sum rel %.1e  u-sum rel %.1e' % (abs(a[0][0] - b[0][0]) / abs(b[0][0]), abs(a[1] - b[1]) / abs(b[1]), abs(a[2] - b[2]) / abs(b[2])))
