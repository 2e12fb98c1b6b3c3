"""Step time of the layer-by-layer route (vn_layered.hip, vn_wide.hip) on config-3 sized inputs (100 000 test functions x 64
points):   python tools/layered_perf.py "128,128,128" ["50,50,50,50,50" ...]
A net the fused kernels cover is run on both routes; a net the tile kernels of vn_wide.hip take (up to 256 wide) is also run on
the GEMM form of the route (VN_LAYERED_NOWIDE=1), shown as route 4/gemms."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
from varnet_amd.engine import VNEngine
n_k, d_in, dim, q, nB = 100000, 3, 2, 64, 14000
n = n_k * q
g = torch.Generator(device='cuda'); g.manual_seed(0)
Input = torch.rand(n, d_in, device='cuda', generator=g) * 2 - 1
gcoef = torch.randn(n, dim, device='cuda', generator=g)
bi = torch.rand(nB, d_in, device='cuda', generator=g) * 2 - 1
bl = torch.randn(nB, device='cuda', generator=g)
rng = np.random.default_rng(0)
N1, dN1 = rng.uniform(0, 1, q), rng.standard_normal(q)
for spec in sys.argv[1:]:
    widths = [int(v) for v in spec.split(',')]
    Fpt = 2 * sum(a * b for a, b in zip([d_in] + widths, widths + [1]))
    in_range = len(widths) <= 6 and max(widths) <= 64
    tiles = max(widths) <= 256
    gemm_leg = tiles and not os.environ.get('VN_PERF_TILES_ONLY')      # counter passes skip the GEMM leg (hundreds of launches, each serialised)
    for kernel in ((0, 4) if in_range else (0,)) + ((40,) if gemm_leg else ()):
        if kernel == 40:
            os.environ['VN_LAYERED_NOWIDE'] = '1'
        e = VNEngine(dim, d_in, widths, True, q, kernel=4 if kernel == 40 else kernel)
        os.environ.pop('VN_LAYERED_NOWIDE', None)
        e.init_params(0); e.set_fe_table(N1, dN1); e.set_interior(0, Input, gcoef, None, n_k=n_k, detJ=1e-6)
        e.set_bic(bi, bl, 9000, 2.0); e.set_weights([1, 1, 1])
        for _ in range(2): e.train_step(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): e.train_step(0)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        print('%-22s route %d%s: %8.2f ms/step  %.3e points/s  algorithmic 6 F_pt -> %.3f of fp32 MFMA peak (%.1f TFLOP/s)'
              % (spec, e.kernel_path()[0], '/gemms' if kernel == 40 else '/tiles' if tiles and e.kernel_path()[0] == 4 else '', ms, n / (ms * 1e-3), (6 * Fpt * n + 3 * Fpt * nB) / (ms * 1e-3) / 157.3e12,
                 (6 * Fpt * n + 3 * Fpt * nB) / (ms * 1e-3) / 1e12), flush=True)
        e.close()
