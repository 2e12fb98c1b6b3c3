"""Longer runs behind tests/test_exact_tables.py (not collected by pytest): the reference's two 1D+t demos with their own
settings, more epochs, then the reference's metric against its tabulated exact values (tests/golden/exact_tables.npz).
    python tests/long_tables.py [epochs_1dt] [epochs_mor]        (on the GPU box; output -> profiles/r3_exact_tables_long.txt)
With epochs_mor >= 100000 (round 5, profiles/r5_explore_mor_120000.txt: ~10 GPU-minutes) the half of the table-pairing question that
the method answers is ASSERTED: for kappa = 0.005 the correct table (cExD4) beats the table the reference's script pairs it with
(cExD3, Operator_1DtMOR.py:216-224) by >= 2 x (measured 3.0 x: 0.042 against 0.129).  For kappa = 0.01/pi the swapped table stays the
closer one (0.103 against 0.085: the trained boundary layer is too diffuse), which is printed, not asserted."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_exact_tables import _mor_errors, _mor_setup, _op1dt_advective, tables, uf  # noqa: E402

e1 = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
e2 = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
inp, d3, d4, kappa = tables()

vn = _op1dt_advective()
np.random.seed(0)
t0 = time.time()
with tempfile.TemporaryDirectory() as tmp:
    res = vn.train(tmp, weight=[10., 10., 1.], smpScheme='optimal', adjustWeight=True, epochNum=e1, saveFreq=5000, verbose=False, lossLag=32)
u = vn.evaluate(x=inp[:, 0:1], t=inp[:, 1:2])
inner = inp[:, 0] <= 0.9
print('Operator_1Dt, kappa = 0.01/pi, [20] net, smpScheme=optimal, adjustWeight: %d epochs in %.0f s, loss %.4e -> %.4e; '
      'l2Err against the table %.5f (points x <= 0.9: %.5f)' % (len(res.lossAll), time.time() - t0, res.lossAll[0], res.lossAll[-1],
                                                                  uf.l2Err(d3, u), uf.l2Err(d3[inner], u[inner])), flush=True)
print('   x      t     table     net')
for (x, t), a, b in zip(inp, d3[:, 0], u[:, 0]):
    print('%6.3f %5.1f %9.5f %9.5f' % (x, t, a, b))
vn.engine.close()

vn = _mor_setup()
np.random.seed(0)
t0 = time.time()
with tempfile.TemporaryDirectory() as tmp:
    res = vn.train(tmp, weight=[10., 10., 1.], smpScheme='uniform', saveMORdata=True, batchNum=20, shuffleData=True,
                   epochNum=e2, saveFreq=2000, verbose=False)
ev = lambda X, k: vn.evaluate(x=X[:, 0:1], t=X[:, 1:2], MORarg=[[k]])
e = _mor_errors(ev, inp, d3, d4, kappa)
print('Operator_1DtMOR, [10,20,30] net, 6 kappa x 20 shuffled mini-batches: %d epochs (%d Adam steps) in %.0f s, loss %.4e -> %.4e; '
      'l2Err against the tables: kappa = 0.01/pi %.5f, kappa = 0.005 %.5f' % (len(res.lossAll), 120 * len(res.lossAll), time.time() - t0,
                                                                             res.lossAll[0], res.lossAll[-1], e[0], e[1]), flush=True)
sw = (uf.l2Err(d4, ev(inp, kappa[0])), uf.l2Err(d3, ev(inp, kappa[1])))
print('the script\'s swapped pairing (Operator_1DtMOR.py:216-224): kappa = 0.01/pi against cExD4 %.5f, kappa = 0.005 against cExD3 %.5f' % sw)
if e2 >= 100000:
    assert 2.0 * e[1] <= sw[1], 'kappa = 0.005: the correct table no longer beats the swapped one by 2 x'
    print('kappa = 0.005: correct pairing beats the swapped one by %.1f x (asserted >= 2 x)' % (sw[1] / e[1]))
vn.engine.close()
