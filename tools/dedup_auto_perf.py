"""
Row-wise against de-duplicated step over problem sizes and nets: the measured basis of the rule behind
`train(dedup='auto')` (varnet_amd/varnet.py::ManageTrainData.dedup_pays).  Run on the GPU box:
    python tools/dedup_auto_perf.py > gpurun_out/dedup_auto_perf.txt
Every line: net, rows per step, unique points, row-wise us/step, de-duplicated us/step, ratio.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def problem_1dt(widths, discNum, tDisc):
    from varnet_amd.domain import Domain1D
    from varnet_amd.adpde import ADPDE
    from varnet_amd.varnet import VarNet
    pde = ADPDE(Domain1D(), diff=0.1 / np.pi, vel=1.0, tInterval=[0, 2.0], IC=lambda x: -np.sin(np.pi * x))
    return VarNet(pde, layerWidth=widths, discNum=discNum, bDiscNum=None, tDiscNum=tDisc)


def problem_2dt(widths, discNum, tDisc):
    from varnet_amd.domain import PolygonDomain2D
    from varnet_amd.adpde import ADPDE
    from varnet_amd.varnet import VarNet
    verts = np.array([[0.0, -0.5], [0.0, -0.2], [0.0, 0.2], [0.0, 0.5], [2.0, 0.5], [2.0, -0.5]])
    BC = [[], [0.0, 1.0, 1.0], [], [], [], []]
    pde = ADPDE(PolygonDomain2D(verts), diff=1e-3, vel=[1., 0.], tInterval=[0, 1.5], BCs=BC, IC=0.0)
    return VarNet(pde, layerWidth=widths, discNum=discNum, bDiscNum=40, tDiscNum=tDisc)


def time_steps(eng, batches, steps):
    import torch
    eng.train_epoch(batches * 5, None)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        eng.train_epoch(batches * steps, None)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (steps * len(batches)))
    return best * 1e6


def run(tag, vn, batchNum=None, steps=200):
    td = vn._build_tdata(batchNum=batchNum)
    td.select_mor(0)
    eng = vn.engine
    eng.set_weights(np.array([1.0, 1.0, 1.0]))
    batches = [td.engine_batch(0, bi) for bi in range(td.batchNum)]
    rows = vn.fixData.nT // td.batchNum
    steps = max(10, min(steps, int(2e8 // max(rows, 1))))
    t_row = time_steps(eng, batches, steps)
    U = td.enable_dedup()
    nmor = vn.fixData.MORbatchNum
    if not U:
        print('%-34s rows/step %9d  de-duplication does not apply: %s' % (tag, rows, getattr(td, 'dedup_reason', '?')))
        eng.close()
        return
    t_dd = time_steps(eng, batches, steps)
    td.disable_dedup()
    print('%-34s rows/step %9d  unique/step %8d  rows/unique %.2f  row-wise %9.1f us  de-dup %9.1f us  ratio %.2f  rule says %s'
          % (tag, rows, U // td.batchNum // nmor, rows * td.batchNum * nmor / U, t_row, t_dd, t_row / t_dd,
             getattr(td, 'dedup_pays', lambda: '?')()), flush=True)
    eng.close()


if __name__ == '__main__':
    which = sys.argv[1:] or ['1dt', '2dt', 'mor']
    if '1dt' in which:
        for widths in ([20], [20] * 3, [10, 20, 30], [50] * 4):
            for disc, td in ((10, 50), (20, 100), (20, 300), (50, 200), (100, 400), (150, 800)):
                run('1D+t %s %dx%d' % (widths, disc, td), problem_1dt(widths, disc, td))
    if '2dt' in which:
        for widths in ([10, 20], [20] * 3, [50] * 5):
            for disc, td in (([8, 6], 8), ([16, 12], 16), ([25, 20], 25), ([50, 40], 50)):
                run('2D+t %s %sx%d' % (widths, disc, td), problem_2dt(widths, disc, td))
    if 'mor' in which:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
        import bench
        for bn in (20, 5, None):
            vn, _ = bench.build_problem(5)
            run('config 5 [10,20,30] batchNum=%s' % bn, vn, batchNum=bn)
