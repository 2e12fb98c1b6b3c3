"""Which route is off?  Loads gpurun_out/fuzz_mismatch.npz (written by tests/fuzz_routes.py when two GPU routes disagree)
and compares every route's gradient with the fp64 oracle and with the oracle run in fp32.
    python tests/diag_fuzz_case.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tf1_graph as og

z = np.load('gpurun_out/fuzz_mismatch.npz', allow_pickle=True)
g = lambda k: (None if z['d_' + k].size == 0 else z['d_' + k])
widths = [int(v) for v in z['widths']]
d_in, dim, q, n_k, bDof = int(z['d_in']), int(z['dim']), int(z['q']), int(z['n_k']), int(z['bDof'])
src, iw, djv, act = bool(z['src']), bool(z['iw']), bool(z['djv']), str(z['act'])
res = {}
for dt, name in ((torch.float64, 'fp64'), (torch.float32, 'fp32')):
    c = (lambda a: None if a is None else a.astype(np.float64 if dt == torch.float64 else np.float32))
    ref, gref = og.loss_and_grad(
        z['flat'].astype(np.float64 if dt == torch.float64 else np.float32), d_in, widths, dt, Input=c(g('Input')), gcoef=c(g('gcoef')),
        source=c(g('source')), N=c(g('N')), dNt=c(g('dNt')), integW=c(g('integW')), intShape=[n_k, q],
        detJ=(c(g('detJ')) if djv else float(g('detJ'))), detJvec=djv, biInput=c(g('biInput')), biLabel=c(g('biLabel')),
        bDof=bDof, biDimVal=2.0, w=g('w'), dim=dim, time_dependent=True, is_source=src, integWflag=iw, activation=act)
    res[name] = (ref['loss'], np.asarray(gref, dtype=np.float64))
l64, g64 = res['fp64']
print('case: %s L=%d widths=%s q=%d n_k=%d   |grad|_inf %.3e  loss %.6e' % (act, len(widths), widths, q, n_k, np.max(np.abs(g64)), l64))
print('fp32 oracle      : grad err %.2e  loss err %.2e' % (np.max(np.abs(res['fp32'][1] - g64)) / np.max(np.abs(g64)), abs(res['fp32'][0] - l64) / abs(l64)))
for route, gr in zip(z['routes'], z['grads']):
    P = g64.size
    e = np.abs(gr[:P] - g64)
    i = int(np.argmax(e))
    print('route %d          : grad err %.2e (param %d of %d, value %.3e ref %.3e)  loss err %.2e'
          % (route, e.max() / np.max(np.abs(g64)), i, P, gr[i], g64[i], abs(gr[P] - l64) / abs(l64)))
