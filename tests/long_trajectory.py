"""Long-run trajectory parity on BASELINE config 1 (Operator_1Dt, 3x20, 96 000 points): N Adam steps on the HIP engine and
on the fp32 oracle (PyTorch-CPU restatement + TF-1 Adam) from the same theta_0; prints the relative loss deviation and
the field difference at checkpoints.  A script, not a collected test (minutes of CPU time):
    python tests/long_trajectory.py [steps]        -> profiles/r2_long_trajectory.txt"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tf1_graph as og
from tests.test_operator_parity_gpu import op1dt, cExact, oracle_kwargs, uf

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
vn = op1dt([20, 20, 20], 20, 300, cEx=cExact)
eng = vn.engine
td = vn._build_tdata(); td.select_mor(0)


class Log:
    verbose = False

    def writeCase(self, s):
        pass


vn.trainRes = Log()
eng.set_weights([1.0, 1.0, 1.0])
trainW, _, _ = vn.trainWeight([10., 10., 1.], td)
eng.set_weights(trainW)
theta0 = eng.get_params()
marks = sorted(set([m for m in (100, 1000, 2000, 3000, 4000, 5000, 10000, 20000, 50000) if m <= steps] + [steps]))
lg = torch.zeros(steps, dtype=torch.float32, device=eng.device)
snaps = {}
t0 = time.perf_counter()
for i in range(steps):
    eng.train_step(0, lg[i:i + 1])
    if i + 1 in marks:
        snaps[i + 1] = eng.get_params()
torch.cuda.synchronize()
t_gpu = time.perf_counter() - t0
gl = lg.cpu().numpy().astype(np.float64)
torch.set_num_threads(16)
kw = oracle_kwargs(vn, td, trainW)
theta = theta0.copy()
adam = og.TF1Adam(theta.size, lr=vn.learning_rate, dtype=np.float32)
cl = np.zeros(steps)
# the same run in fp64: how far do two fp32 implementations sit from it (and so from each other) by themselves?
kw64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v) for k, v in kw.items()}
th64 = theta0.astype(np.float64)
adam64 = og.TF1Adam(th64.size, lr=vn.learning_rate, dtype=np.float64)
dl = np.zeros(steps)
ui = vn.fixData.uniform_input
print('config 1 (Operator_1Dt, 3x20, 96 000 points), %d TF-1 Adam steps; HIP engine %.1f s' % (steps, t_gpu), flush=True)
print('%8s %14s %14s %12s %14s %14s | vs the fp64 run, max so far: %10s %10s' % ('step', 'loss hip', 'loss oracle', 'rel dev', 'max dev so far', 'field l2 diff', 'hip', 'fp32 oracle'), flush=True)
t0 = time.perf_counter()
for i in range(steps):
    res, g = og.loss_and_grad(theta, vn.inpDim, vn.layerWidth, torch.float32, **kw)
    theta = adam.step(theta, g)
    cl[i] = res['loss']
    res64, g64 = og.loss_and_grad(th64, vn.inpDim, vn.layerWidth, torch.float64, **kw64)
    th64 = adam64.step(th64, g64)
    dl[i] = res64['loss']
    if (i + 1) % 250 == 0:
        print('  ... oracle step %d (%.0f s)' % (i + 1, time.perf_counter() - t0), file=sys.stderr, flush=True)
    if i + 1 in marks:
        dev = np.abs(gl[:i + 1] - cl[:i + 1]) / np.abs(cl[:i + 1])
        u_g = og.forward(snaps[i + 1].astype(np.float64), 2, [20, 20, 20], torch.float64, ui)
        u_c = og.forward(theta.astype(np.float64), 2, [20, 20, 20], torch.float64, ui)
        d_h = (np.abs(gl[:i + 1] - dl[:i + 1]) / np.abs(dl[:i + 1])).max()
        d_c = (np.abs(cl[:i + 1] - dl[:i + 1]) / np.abs(dl[:i + 1])).max()
        print('%8d %14.6e %14.6e %12.2e %14.2e %14.2e | %33.2e %10.2e' % (i + 1, gl[i], cl[i], dev[-1], dev.max(), uf.l2Err(u_c, u_g), d_h, d_c), flush=True)
print('oracle %.1f s on 16 threads' % (time.perf_counter() - t0))
