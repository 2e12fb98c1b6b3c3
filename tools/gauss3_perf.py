"""Three-point Gauss in 2D+t (integPnum = 3: 216 quadrature points per test function, the integW path of TFModel.py:660) on the
config-3 problem and net (5x50), 30 000 test functions x 216 = 6.48 M rows per step: the row-wise step (two-pass fused route:
a test function does not fit one 128-point tile, 8 F_pt per ROW) against train(dedup=True) (one evaluation per unique quadrature
point: 27 per element, 8 F_pt per UNIQUE point).       python tools/gauss3_perf.py [steps]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from varnet_amd import ADPDE, PolygonDomain2D, VarNet

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
verts = np.array([[0.0, -0.5], [0.0, -0.2], [0.0, 0.2], [0.0, 0.5], [2.0, 0.5], [2.0, -0.5]])
BC = [[], [0.0, 1.0, 1.0], [], [], [], []]
pde = ADPDE(PolygonDomain2D(verts), diff=1e-3, vel=[1., 0.], tInterval=[0, 1.5], BCs=BC, IC=0.0)
vn = VarNet(pde, layerWidth=[50] * 5, discNum=[50, 40], bDiscNum=40, tDiscNum=15, integPnum=3)
fd, eng = vn.fixData, vn.engine
td = vn._build_tdata()
td.select_mor(0)
eng.set_weights(np.array([1.0, 1.0, 1.0]))
gb = eng.bind_grad_buffer()
print('test functions %d x %d points = %d rows, route %s' % (fd.nt, fd.integNum, fd.nT, eng.kernel_path()))
F_pt = 2 * (3 * 50 + 4 * 2500 + 50)


def run(label, units, fl):
    eng.init_params(seed=0)
    eng.train_epoch([0] * 3, None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.train_epoch([0] * steps, None)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print('%-28s %7.3f ms/step  %.3e rows/s  loss after %d steps %.6e  (%.1f TFLOP/s of the %s)'
          % (label, dt * 1e3, fd.nT / dt, steps + 3, float(gb[eng.P].item()), fl * units / dt / 1e12, 'formulation run'))
    return dt


a = run('row-wise (two-pass fused)', fd.nT, 8 * F_pt)
U = td.enable_dedup()
print('unique points %d (%.2f rows per point)' % (U, fd.nT / max(U, 1)))
b = run('de-duplicated', U, 8 * F_pt)
print('speed-up %.2fx' % (a / b))
eng.close()
