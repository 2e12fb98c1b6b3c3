"""
The reference's 1D+t demo (/root/reference/Operator_1Dt.py:69-186) on the MI355X engine: same PDE, same network
([20]), same discretisation (discNum=20, tDiscNum=300 -> 96 000 training points), same call sequence -- the only
edit a user of the reference makes is the import line (varnet_amd instead of the flat VarNet modules).

    python examples/operator_1dt.py [out_folder] [epochs] [smpScheme] [lossLag]

The reference runs `train(..., smpScheme='optimal', adjustWeight=True)` until `loss < tol` or 500 000 epochs; pass an
epoch count to bound the run.  Prints the script's own acceptance metric, "Normalized approximation error".
"""
import os
import sys
import time

import numpy as np
from numpy import pi, sin

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varnet_amd import ADPDE, Domain1D, VarNet, UF          # reference: from Domain import Domain1D; from ADPDE import ADPDE; ...

uf = UF()
u, D, T = 1.0, 0.1 / pi, 2.0


def IC(x):
    return -sin(pi * x)


def cExact(x, t, trunc=800):
    """Fourier-series solution (Operator_1Dt.py:78-108)."""
    p = np.arange(0, trunc + 1.0).reshape(1, trunc + 1)
    c0 = 16 * pi ** 2 * D ** 3 * u * np.exp(u / D / 2 * (x - u * t / 2))
    c1n = (-1) ** p * 2 * p * np.sin(p * pi * x) * np.exp(-D * p ** 2 * pi ** 2 * t)
    c1d = u ** 4 + 8 * (u * pi * D) ** 2 * (p ** 2 + 1) + 16 * (pi * D) ** 4 * (p ** 2 - 1) ** 2
    c1 = np.sinh(u / D / 2) * np.sum(c1n / c1d, axis=-1, keepdims=True)
    c2n = (-1) ** p * (2 * p + 1) * np.cos((p + 0.5) * pi * x) * np.exp(-D * (2 * p + 1) ** 2 * pi ** 2 * t / 4)
    c2d = u ** 4 + (u * pi * D) ** 2 * (8 * p ** 2 + 8 * p + 10) + (pi * D) ** 4 * (4 * p ** 2 + 4 * p - 3) ** 2
    c2 = np.cosh(u / D / 2) * np.sum(c2n / c2d, axis=-1, keepdims=True)
    c = c0 * (c1 + c2)
    if np.size(t) > 1:
        ind0 = t == 0
        c[ind0] = IC(x[ind0])
    elif t == 0:
        c = IC(x)
    return c


def main():
    folder = sys.argv[1] if len(sys.argv) > 1 else 'out_operator_1dt'
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    scheme = sys.argv[3] if len(sys.argv) > 3 else 'optimal'
    lag = int(sys.argv[4]) if len(sys.argv) > 4 else None              # lossLag: read-back schedule (None = train()'s default: 8, exact; 0 = one read-back per epoch)
    domain = Domain1D()
    pde = ADPDE(domain, diff=D, vel=u, timeDependent=True, tInterval=[0, T], IC=IC, cEx=cExact)
    vn = VarNet(pde, layerWidth=[20], discNum=20, bDiscNum=None, tDiscNum=300, processors='GPU:0')
    os.makedirs(folder, exist_ok=True)
    t0 = time.perf_counter()
    vn.train(folder, weight=[1.e1, 1.e1, 1.], smpScheme=scheme, adjustWeight=True, epochNum=epochs, saveFreq=1000,
             verbose=False, lossLag=lag)
    dt = time.perf_counter() - t0
    vn.loadModel()
    sim = vn.simRes()
    cEx, cApp = vn.fixData.cEx, vn.evaluate()
    s = '\n==========================================================\nSimulation results:\n\n'
    s += 'Normalized approximation error: %2.5f' % uf.l2Err(cEx, cApp)
    print(s)
    vn.trainRes.writeComment(s)
    print('%d epochs in %.1f s (%.1f us/epoch incl. monitors every 1000), kernel path %s, snapshot errors %s'
          % (len(vn.trainRes.lossAll), dt, dt / max(len(vn.trainRes.lossAll), 1) * 1e6, vn.engine.kernel_path(),
             ['%.4f' % e for e in sim['l2Err']]))


if __name__ == '__main__':
    main()
