import os, sys, numpy as np, torch
os.environ['VN_FULL_GRID'] = '1'
sys.path.insert(0, '.')
from varnet_amd.engine import VNEngine
for widths, d_in, dim, q in (([10, 20, 30], 3, 1, 16), ([50] * 5, 3, 2, 64)):
    e = VNEngine(dim, d_in, widths, True, q)
    e.init_params(0); e.set_fe_table(np.ones(q), np.ones(q))
    X = torch.rand(1, d_in, device='cuda'); G = torch.randn(1, dim, device='cuda')
    e.set_interior(0, X[:0], G[:0], None, n_k=0, detJ=1e-3); e.set_bic(None, None, 0, 1.0); e.set_weights([1, 1, 1])
    for _ in range(5): e.train_step(0)
    torch.cuda.synchronize()
    st = e.debug_stamps()
    t0 = st[0]
    print(widths, 'cycles from kernel entry: loads issued %d | zero-fill + barrier %d | scatter done %d | prologue end (all loads waited) %d | set-up end %d | flush end %d | store end %d'
          % (st[5] - t0, st[6] - t0, st[7] - t0, st[1] - t0, st[2] - t0, st[3] - t0, st[4] - t0))
    e.close()
