"""
The reference's 2D+t demo (/root/reference/Operator_2Dt.py:78-186) on the MI355X engine with the import line swapped:
rectangle [0,2]x[-.5,.5], kappa = 1e-3, v = (1,0), inlet c = 1 on x = 0, |y| < 0.2, T = 1.5; VarNet(layerWidth=[10,20],
discNum=[80,40], bDiscNum=40, tDiscNum=75) -> 240 000 test functions x 64 = 15.36 M training points per epoch;
train(weight=[5,1,1], smpScheme='uniform').  Prints the script's "approximation error" against its analytical solution
(Leij & Dane, integrated over time) at t = T.

    python examples/operator_2dt.py [out_folder] [epochs] [rowwise]
Without a third argument train() chooses the formulation itself (dedup='auto', the default): on this uniform, unshuffled set
that is the de-duplicated one -- one network evaluation per unique quadrature point, 2.05 M points instead of 15.36 M rows;
same loss and gradient up to fp32 rounding.  `rowwise` as third argument asks for train(dedup=False), the reference's one
evaluation per (test function, point) row.
"""
import os
import sys
import time

import numpy as np
from numpy import exp, pi
from scipy.special import erf

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varnet_amd import ADPDE, PolygonDomain2D, VarNet, UF     # reference: from Domain import PolygonDomain2D; ...

uf = UF()
T, q, kappa, c0, a, nt = 1.5, [1., 0.], 1.e-3, 1.0, 0.2, 151


def cExFun(x, t=None):
    """Analytical solution, integrated over the temporal coordinate (Operator_2Dt.py:89-132) -> [nx, nt] or [nx, len(t)]."""
    nx = len(x)
    tcoord = np.linspace(0, T, num=nt)[1:].reshape(-1, 1)
    dt = T / (nt - 1)
    Input = uf.pairMats(x, tcoord)
    x1, x2, tc = Input[:, 0:1], Input[:, 1:2], Input[:, 2:3]
    integ = c0 * x1 / 4 / np.sqrt(pi * kappa) * tc ** (-1.5) * exp(-(x1 - q[0] * tc) ** 2 / 4 / kappa / tc)
    denom = 1 / 2 / np.sqrt(kappa * tc)
    integ = integ * (erf((a + x2) * denom) + erf((a - x2) * denom))
    integ = dt * np.cumsum(integ.reshape(nx, nt - 1), axis=1)
    integ = np.hstack([np.zeros([nx, 1]), integ])
    if t is not None:
        integ = integ[:, uf.nodeNum(np.linspace(0, T, num=nt), t)]
    ind = (x[:, 0] < 1.e-4) * (x[:, 1] < a) * (-a < x[:, 1])
    integ[ind, :] = c0
    return integ


def main():
    folder = sys.argv[1] if len(sys.argv) > 1 else 'out_operator_2dt'
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    dedup = False if (len(sys.argv) > 3 and sys.argv[3] == 'rowwise') else 'auto'
    vertices = np.array([[0.0, -0.5], [0.0, -a], [0.0, a], [0.0, 0.5], [2.0, 0.5], [2.0, -0.5]])
    domain = PolygonDomain2D(vertices)
    BC = [[], [0.0, 1.0, c0], [], [], [], []]                  # Dirichlet c = c0 on the inlet segment, natural elsewhere
    pde = ADPDE(domain, diff=kappa, vel=q, tInterval=[0, T], BCs=BC, IC=0.0)
    vn = VarNet(pde, layerWidth=[10, 20], discNum=[80, 40], bDiscNum=40, tDiscNum=75, processors='GPU:0')
    fd = vn.fixData
    print('test functions %d, training points per epoch %d, BC/IC points %d' % (fd.nt, fd.nT, int(np.sum(fd.biDof))))
    os.makedirs(folder, exist_ok=True)
    t0 = time.perf_counter()
    vn.train(folder, weight=[5., 1., 1.], smpScheme='uniform', epochNum=epochs, saveFreq=500, verbose=False, dedup=dedup)
    dt = time.perf_counter() - t0
    vn.loadModel()
    coord = domain.getMesh(discNum=[60, 30], bDiscNum=20).coordinates
    cEx = cExFun(coord, [T])
    cApp = vn.evaluate(coord, T)
    print('%s%d epochs in %.1f s (%.2f ms/epoch, %.3e training points/s); approximation error at t = T: %2.5f'
          % ('de-duplicated formulation (%d unique points): ' % vn.dedup_state['unique_points'] if vn.dedup_state['on']
             else 'row-wise formulation (%s): ' % vn.dedup_state['reason'], len(vn.trainRes.lossAll), dt, dt / len(vn.trainRes.lossAll) * 1e3, fd.nT * len(vn.trainRes.lossAll) / dt,
             uf.l2Err(cEx, cApp)))


if __name__ == '__main__':
    main()
