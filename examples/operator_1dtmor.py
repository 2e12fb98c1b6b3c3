"""
The reference's parametric (model-order-reduction) demo (/root/reference/Operator_1DtMOR.py:163-210) on the MI355X engine:
1D+t advection-diffusion with the diffusivity as a third network input, 6 log-spaced values in [0.003, 0.033];
VarNet(layerWidth=[10,20,30], discNum=150, bDiscNum=75, tDiscNum=800) -> 120 000 test functions x 16 points per kappa;
train(weight=[10,10,1], smpScheme='uniform', saveMORdata=True, batchNum=20, shuffleData=True): 6 x 20 Adam steps per epoch.

    python examples/operator_1dtmor.py [out_folder] [epochs]
"""
import os
import sys
import time

import numpy as np
from numpy import pi, sin

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varnet_amd import ADPDE, Domain1D, MOR, VarNet, UF

uf = UF()
u, T = 1.0, 2.0


def IC(x):
    return -sin(pi * x)


def diffFun(x, t=0, D=0.01):
    return D * np.ones([np.shape(x)[0], 1])


def discDiff(discNum=6):
    return np.array([0.003 * (11 ** (n / (discNum - 1))) for n in range(discNum)])[np.newaxis].T


def main():
    folder = sys.argv[1] if len(sys.argv) > 1 else 'out_operator_1dtmor'
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    MORvar = MOR(diffFun, ['D'], [[0.003, 0.033]])
    domain = Domain1D()
    pde = ADPDE(domain, diff=diffFun, vel=u, timeDependent=True, tInterval=[0, T], IC=IC, MORvar=MORvar)
    vn = VarNet(pde, layerWidth=[10, 20, 30], discNum=150, bDiscNum=75, tDiscNum=800, MORdiscScheme=discDiff, processors='GPU:0')
    fd = vn.fixData
    os.makedirs(folder, exist_ok=True)
    t0 = time.perf_counter()
    vn.train(folder, weight=[1.e1, 1.e1, 1.], smpScheme='uniform', saveMORdata=True, batchNum=20, shuffleData=True,
             epochNum=epochs, saveFreq=50, verbose=False)
    dt = time.perf_counter() - t0
    vn.loadModel()
    n = len(vn.trainRes.lossAll)
    r64, _, _, _ = vn.residual(fp64=True)
    r32, _, _, _ = vn.residual()
    print('%d epochs (%d Adam steps) in %.1f s: %.1f ms/epoch, %.3e training points/s; loss %.1f -> %.1f; PDE residual norm '
          'fp64 %.6f / fp32 %.6f (averaged over the %d kappa batches)'
          % (n, vn.engine.step, dt, dt / n * 1e3, fd.nT * fd.MORbatchNum * n / dt, vn.trainRes.lossAll[0], vn.trainRes.lossAll[-1],
             r64, r32, fd.MORbatchNum))
    for b in (0, 5):
        c = vn.evaluate(batch=b)
        print('  kappa batch %d: model range [%.3f, %.3f] on uniform_input' % (b, c.min(), c.max()))


if __name__ == '__main__':
    main()
