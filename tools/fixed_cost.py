"""Fixed cost of one fused launch: time of vn_grad on an EMPTY batch (n_k = 0, nB = 0: prologue + epilogue only)
and on 1 / 2 tiles per workgroup, for the 5x50 and 4x50 nets.   python tools/fixed_cost.py"""
import os, sys, numpy as np, torch
os.environ['VN_FULL_GRID'] = '1'   # fixed cost of a full 256-workgroup launch, also on the empty batch
sys.path.insert(0, '.')
from varnet_amd.engine import VNEngine
for widths, d_in, dim, q in (([50] * 5, 3, 2, 64), ([50] * 4, 2, 1, 16), ([10, 20, 30], 3, 1, 16)):
    for tiles_per_wg in (0, 1, 2, 4):
        n_k = tiles_per_wg * 256 * (128 // q)
        n = n_k * q
        e = VNEngine(dim, d_in, widths, True, q)
        e.init_params(0)
        rng = np.random.default_rng(0)
        e.set_fe_table(rng.uniform(0, 1, q), rng.standard_normal(q))
        X = torch.rand(max(n, 1), d_in, device='cuda'); G = torch.randn(max(n, 1), dim, device='cuda')
        e.set_interior(0, X[:n], G[:n], None, n_k=n_k, detJ=1e-3)
        e.set_bic(None, None, 0, 1.0)
        e.set_weights([1, 1, 1])
        for _ in range(5): e.train_step(0)
        torch.cuda.synchronize()
        e.profile_begin()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(200): e.train_step(0)
        ev1.record(); torch.cuda.synchronize()
        ms, nl, kn = e.profile_end()
        print('net %s q=%d: %d tiles/WG: fused kernel %.1f us, whole step %.1f us' % (widths, q, tiles_per_wg, ms * 1e3, ev0.elapsed_time(ev1) / 200 * 1e3))
        e.close()

# with a -DVN_FIXSTAMPS build (VARNET_HIP_LIB=.../libvarnet_hip_fix.so): where the empty launch's cycles go
import os
if 'fix' in os.environ.get('VARNET_HIP_LIB', ''):
    for widths, d_in, dim, q in (([50] * 5, 3, 2, 64), ([10, 20, 30], 3, 1, 16), ([20] * 3, 2, 1, 16)):
        e = VNEngine(dim, d_in, widths, True, q)
        e.init_params(0); e.set_fe_table(np.ones(q), np.ones(q))
        X = torch.rand(1, d_in, device='cuda'); G = torch.randn(1, dim, device='cuda')
        e.set_interior(0, X[:0], G[:0], None, n_k=0, detJ=1e-3); e.set_bic(None, None, 0, 1.0); e.set_weights([1, 1, 1])
        for _ in range(3): e.train_step(0)
        torch.cuda.synchronize()
        st = e.debug_stamps()
        d = [st[i + 1] - st[i] for i in range(4)]
        # s_memtime counts shader cycles (MI355X_MICROARCH.md): 2.4 GHz nominal
        print('empty %s launch, workgroup 0 wave 0 (s_memtime cycles; us at 2.4 GHz): prologue %d (%.2f)  set-up %d (%.2f)  flush %d (%.2f)  store + loss partials %d (%.2f)'
              % ((widths,) + tuple(v for x in d for v in (x, x / 2400.0))))
        e.close()
