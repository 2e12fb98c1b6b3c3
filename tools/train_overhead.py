"""What a reference caller gets from train() with its DEFAULT arguments (saveFreq=100: a monitor -- residual, loss split with the loss
field, checkpoint when the loss improved, result files -- every 100 epochs) against the bare step rate, on a BASELINE config:
    python tools/train_overhead.py [config 3|2|1] [epochs] [saveFreq]
Prints the wall time of train(), the epochs' own time (vn_train_epoch on the same formulation) and, with cProfile, where the host spends
the difference."""
import cProfile
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
kw = {}
if len(sys.argv) > 3:
    kw['saveFreq'] = int(sys.argv[3])
vn, name = bench.build_problem(cfg)
np.random.seed(0)
pr = cProfile.Profile()
with tempfile.TemporaryDirectory() as tmp:
    t0 = time.perf_counter()
    pr.enable()
    res = vn.train(tmp, weight=[5., 1., 1.] if cfg == 3 else [10., 10., 1.], epochNum=epochs, verbose=False, **kw)
    pr.disable()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
eng, td = vn.engine, vn.tData
t1 = time.perf_counter()
eng.train_epoch([0] * 200, None)
torch.cuda.synchronize()
step = (time.perf_counter() - t1) / 200
print('%s' % name)
print('train(): %d epochs in %.2f s = %.3f ms/epoch; formulation: %s; bare step %.3f ms -> the steps are %.1f %% of train()'
      % (len(res.lossAll), dt, dt / len(res.lossAll) * 1e3, vn.dedup_state, step * 1e3, 100 * step * len(res.lossAll) / dt))
if vn.dedup_state.get('on'):
    # a monitor's loss split (splitLoss -> vn_eval_loss with the loss field) on the de-duplicated formulation and row-wise
    def tsplit():
        vn.splitLoss(td); torch.cuda.synchronize(); t2 = time.perf_counter()
        for _ in range(5):
            vn.splitLoss(td)
        torch.cuda.synchronize()
        return (time.perf_counter() - t2) / 5
    a = tsplit()
    eng.debug_point_route(8)
    b = tsplit()
    eng.debug_point_route(0)
    print('splitLoss of a monitor: %.2f ms on the de-duplicated formulation, %.2f ms row-wise (vn_debug_point_route | 8)' % (a * 1e3, b * 1e3))
st = pstats.Stats(pr)
st.sort_stats('cumulative')
st.print_stats(14)
st.sort_stats('tottime')
st.print_stats(16)
