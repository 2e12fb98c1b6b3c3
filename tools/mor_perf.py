"""Epoch time on a BASELINE-cfg-5-like MOR workload (6 diffusivities x 20 mini-batches, [10,20,30] net):
one host call per pass (vn_train_epoch) vs one per mini-batch."""
import sys, time, tempfile, numpy as np, torch
from math import pi
sys.path.insert(0, '.')
from varnet_amd.domain import Domain1D
from varnet_amd.adpde import ADPDE
from varnet_amd.mor import MOR
from varnet_amd.varnet import VarNet
def diffFun(x, t=0, D=0.01): return D * np.ones([np.shape(x)[0], 1])
def disc(discNum=6): return np.array([0.003 * (11 ** (n / (discNum - 1))) for n in range(discNum)])[np.newaxis].T
mor = MOR(diffFun, ['D'], [[0.003, 0.033]])
pde = ADPDE(Domain1D(), diff=diffFun, vel=1.0, timeDependent=True, tInterval=[0, 2.0], IC=lambda x: -np.sin(pi * x), MORvar=mor)
vn = VarNet(pde, layerWidth=[10, 20, 30], discNum=150, bDiscNum=None, tDiscNum=800, MORdiscScheme=disc)
fd, eng = vn.fixData, vn.engine
td = vn._build_tdata(batchNum=20)
eng.set_weights([1.0, 1.0, 1.0])
acc = torch.zeros((), device='cuda')
def epoch():
    for mb in range(fd.MORbatchNum):
        td.select_mor(mb)
        vn.optimIter(td, mb, acc)
for mode in ('train_epoch', 'per-step'):
    if mode == 'per-step': vn.world = 2; vn.dist = None; vn._allreduce = lambda gb: None    # force the per-mini-batch host loop
    for _ in range(2): epoch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): epoch()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    steps = fd.MORbatchNum * td.batchNum
    print('%-12s %.2f ms/epoch (%d steps, %.1f us/step)  %.3e points/s' % (mode, dt * 1e3, steps, dt / steps * 1e6, fd.nT * fd.MORbatchNum / dt))
