"""Diagnostic (not collected by pytest): per-layer gradient errors of the AUTO kernel against the oracle.
   python tests/diag_fused.py [1]   -- lives under tests/ because only tests may use oracle/."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_engine_gpu import synth, make_engine, oracle_eval
from oracle import tf1_graph as og
def run(d_in, dim, widths, q, n_k, nB, bDof, source=False, integW=False, detJvec=False):
    d = synth(1, d_in, dim, widths, q, n_k, nB, bDof, source, integW, detJvec)
    eng = make_engine(d_in, dim, widths, q, source, integW, 0)
    eng.init_params(seed=3)
    flat = eng.get_params() + 0.05*np.random.default_rng(5).standard_normal(eng.P).astype(np.float32)
    eng.set_params(flat)
    eng.set_fe_table(d['N1'], d['dNt1'], d['integW'])
    eng.set_interior(0, d['Input'], d['gcoef'], d['source'], n_k=n_k, detJ=d['detJ'])
    eng.set_bic(d['biInput'], d['biLabel'], bDof, 2.0)
    eng.set_weights(d['w'])
    ref, gref = oracle_eval(flat, d, d_in, dim, widths, q, n_k, bDof, source, integW, detJvec)
    gb = eng.bind_grad_buffer(); eng.grad(0); torch.cuda.synchronize()
    g = gb.cpu().numpy()
    print(widths, 'd_in', d_in, 'dim', dim, 'q', q, 'loss', g[eng.P], ref['loss'])
    off = 0
    for (i, o) in og.layer_dims(d_in, widths):
        gw, rw = g[off:off+i*o], gref[off:off+i*o]; off += i*o
        gbb, rb = g[off:off+o], gref[off:off+o]; off += o
        print('  layer %dx%d  W err %.2e (max ref %.2e)  b err %.2e (max ref %.2e)' % (i, o, abs(gw-rw).max(), abs(rw).max(), abs(gbb-rb).max(), abs(rb).max()))
        if len(sys.argv) > 1 and abs(gw-rw).max() > 1e-3*abs(rw).max():
            E = abs(gw-rw).reshape(i, o) > 1e-3*abs(rw).max()
            print('    bad rows (in-features):', np.nonzero(E.any(axis=1))[0].tolist())
            print('    bad cols (out-features):', np.nonzero(E.any(axis=0))[0].tolist())
    eng.close()
if len(sys.argv) > 1:
    run(3, 2, [50, 50, 50, 50, 50], 64, 9, 77, 40)
else:
    run(3, 1, [10, 20, 30], 16, 21, 19, 7)
    run(3, 1, [30, 30, 30], 16, 21, 19, 7)
    run(3, 1, [20, 20, 20], 16, 21, 19, 7)
    run(3, 2, [30, 30, 30], 16, 21, 19, 7)
    run(2, 1, [30, 30], 16, 21, 19, 7)
