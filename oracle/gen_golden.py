"""
ORACLE tooling (test infrastructure): generate the golden fixtures in tests/golden from the
reference's own, directly importable NumPy modules (FiniteElement, Domain, UtilityFunc, ADPDE,
MOR under /root/reference).  Runs ONLY in the build container -- the reference never travels
to the GPU box; the small .npz outputs are committed.

    MPLBACKEND=Agg python oracle/gen_golden.py

Only modules that import unmodified are used.  `VarNet.py`, `VarNetUtility.py` and `TFModel.py`
need TensorFlow 1.10 (absent, ordinary ModuleNotFoundError) and are NOT imported, stubbed or
partially executed; the data-assembly code in them (trainingPoints, biTrainPoints, gcoef ...)
is pinned through its ingredients below plus the values SURVEY.md 8(c) records from the survey
session (tests/test_problem_layer.py::test_survey_known_answers).
"""
import os
import sys

sys.dont_write_bytecode = True      # importing the reference must not write __pycache__ into /root/reference (read-only input)

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def main():
    os.environ.setdefault('MPLBACKEND', 'Agg')
    sys.path.insert(0, REF)
    import FiniteElement as RFE
    import Domain as RD
    import UtilityFunc as RU
    import ADPDE as RA
    import MOR as RM
    os.makedirs(OUT, exist_ok=True)
    uf = RU.UF()

    # (1) FE tables + (2) basisTot
    fe = {}
    for D, ip in [(1, 2), (2, 2), (3, 2), (1, 3), (2, 3), (3, 3)]:
        f = RFE.FE(D, ip)
        k = 'D%d_ip%d_' % (D, ip)
        fe[k + 'basMultiInd'] = f.basMultiInd
        fe[k + 'IntegP'] = f.IntegP
        fe[k + 'basVal'] = f.basVal
        fe[k + 'basDeriVal'] = f.basDeriVal
        fe[k + 'elemCoord'] = f.elemCoord
        fe[k + 'delta'] = f.delta
        fe[k + 'IntegW'] = np.zeros(0) if f.IntegW is None else f.IntegW
        fe[k + 'massVec'], fe[k + 'massDelta'] = f.massVec, f.massDelta      # FiniteElement.py:121-123, 438-499
        hVec = np.array([[0.3], [0.05], [0.7]])[:D]
        integNum, nT, detJ, delta, iw, N, dN = f.basisTot(3, hVec)
        fe[k + 'bt_hVec'] = hVec
        fe[k + 'bt_scalars'] = np.array([integNum, nT, detJ])
        fe[k + 'bt_delta'] = delta
        fe[k + 'bt_intWeight'] = np.zeros(0) if iw is None else iw
        fe[k + 'bt_N'] = N
        fe[k + 'bt_dN'] = dN
    np.savez_compressed(os.path.join(OUT, 'fe_tables.npz'), **fe)

    # (3) meshes
    ms = {}
    m = RD.Domain1D().getMesh(20)
    ms['d1_coord'], ms['d1_he'], ms['d1_bdof'], ms['d1_bcoord'] = m.coordinates, m.he, m.bdof, m.bCoordinates
    m = RD.Domain1D(np.array([0.5, 3.0])).getMesh(7)
    ms['d1b_coord'], ms['d1b_he'] = m.coordinates, m.he
    verts = np.array([[0.0, -0.5], [0.0, -0.2], [0.0, 0.2], [0.0, 0.5], [2.0, 0.5], [2.0, -0.5]])
    dom = RD.PolygonDomain2D(verts)
    m = dom.getMesh([8, 4], 4)
    ms['p2_vertices'] = verts
    ms['p2_coord'], ms['p2_he'], ms['p2_bdof'] = m.coordinates, m.he, np.array(m.bdof)
    ms['p2_bcoord'] = np.vstack(m.bCoordinates)
    ms['p2_measure'] = np.array(dom.measure)
    ms['p2_lim'] = dom.lim
    ms['p2_bgeom'] = dom.boundryGeom
    obs = [np.array([[0.5, -0.2], [1.0, -0.2], [1.0, 0.2], [0.5, 0.2]])]
    dom = RD.PolygonDomain2D(verts, obs)
    m = dom.getMesh([16, 9], 3)
    ms['p2o_obs'] = obs[0]
    ms['p2o_coord'], ms['p2o_he'], ms['p2o_bdof'] = m.coordinates, m.he, np.array(m.bdof)
    ms['p2o_bcoord'] = np.vstack(m.bCoordinates)
    ms['p2o_bIndNum'] = np.array(m.bIndNum)
    pts = np.array([[0.1, 0.0], [0.7, 0.0], [2.5, 0.0], [1.5, 0.4], [0.0, 0.0]])
    ms['p2o_pts'] = pts
    ms['p2o_inside'] = dom.isInside(pts)
    ms['p2_scale'] = dom.scaleCoord(pts)
    np.savez_compressed(os.path.join(OUT, 'meshes.npz'), **ms)

    # (4) utilities
    ut = {}
    a = np.arange(6.0).reshape(3, 2)
    b = np.array([[10.0], [20.0]])
    ut['pm_a'], ut['pm_b'] = a, b
    ut['pm_ab'] = uf.pairMats(a, b)
    ut['pm_ab_rev'] = uf.pairMats(a, b, reverse=True)
    x = np.array([1.0, 2.0, 3.5, -1.0]); y = np.array([1.1, 1.9, 3.0, -0.5])
    ut['l2_x'], ut['l2_y'], ut['l2'] = x, y, np.array(uf.l2Err(x, y))
    ut['poly'] = verts
    ut['polyArea'] = np.array(uf.polyArea(verts))
    np.savez_compressed(os.path.join(OUT, 'utility.npz'), **ut)

    # (5) ADPDE BC normalisation
    BC = [[], [0.0, 1.0, 1.0], [], [1.0, 0.0, 2.0], [1.0, 2.0, 0.5], []]
    pde = RA.ADPDE(RD.PolygonDomain2D(verts), diff=1e-3, vel=[1., 0.], tInterval=[0, 1.5], BCs=BC, IC=0.0)
    xs = np.array([[0.0, 0.1], [0.3, -0.2], [1.0, 0.4]])
    ts = np.array([[0.1], [0.2], [0.3]])
    ad = {'x': xs, 't': ts}
    ad['BCtype'] = np.array(pde.BCtype)
    ad['BCab'] = np.array([[bc[0], bc[1]] for bc in pde.BCs], dtype=float)
    ad['BCg'] = np.hstack([bc[2](xs, ts) for bc in pde.BCs])
    ad['diff'] = pde.diffFun(xs, ts); ad['vel'] = pde.velFun(xs, ts)
    ad['source'] = pde.sourceFun(xs, ts); ad['d_diff'] = pde.d_diffFun(xs, ts)
    ad['IC'] = pde.IC(xs)
    ad['timeDependent'] = np.array(pde.timeDependent)
    np.savez_compressed(os.path.join(OUT, 'adpde.npz'), **ad)

    # (6) MOR discretisation of the Operator_1DtMOR setup (+ a two-function case)
    def diffFun(x, t=0, D=0.01):
        return D * np.ones([np.shape(x)[0], 1])

    def discDiff(discNum=6):
        return np.array([0.003 * (11 ** (n / (discNum - 1))) for n in range(discNum)])[np.newaxis].T

    mor = RM.MOR(diffFun, ['D'], [[0.003, 0.033]])
    da = mor.discretizeArg(discDiff)
    mo = {'m1_disc0': da[0], 'm1_argInd': mor.argIndex(da), 'm1_varNum': np.array(mor.varNum)}

    def velFun(x, t=0, a=1.0, b=2.0):
        return a * np.ones([np.shape(x)[0], 1])

    mor2 = RM.MOR([diffFun, velFun], [['D'], ['b', 'a']], [[[0.1, 0.2]], [[3.0, 4.0], [1.0, 2.0]]])
    da2 = mor2.discretizeArg([3, [2, 4]])
    mo['m2_disc0'], mo['m2_disc1'] = da2[0], da2[1]
    mo['m2_argInd'] = mor2.argIndex(da2)
    mo['m2_names1'] = np.array(mor2.ArgNames[1])
    np.savez_compressed(os.path.join(OUT, 'mor.npz'), **mo)

    # (8b) rejection sampling (UtilityFunc.py:342-404) from a fixed NumPy seed: one segment, and two segments with dofT
    rs = {}

    def dens(x=None):
        if x is None:
            x = grid
        return np.exp(-8.0 * (x[:, 0:1] - 0.3) ** 2) + 0.05

    grid = np.linspace(-1, 1, 41).reshape(-1, 1)
    np.random.seed(2024)
    rs['one'] = uf.rejectionSampling(dens, lambda: np.random.uniform(-1, 1, [50, 1]), 37)
    grid = np.vstack([np.linspace(-1, 1, 30).reshape(-1, 1), np.linspace(-1, 1, 20).reshape(-1, 1)])
    np.random.seed(7)
    rs['two'] = uf.rejectionSampling(dens, lambda: np.random.uniform(-1, 1, [50, 1]), [11, 9], [30, 20])
    np.savez_compressed(os.path.join(OUT, 'rejection.npz'), **rs)

    # (9) ContourPlot: plotting grid and the field arrays conPlot / snap1Dt draw (ContourPlot.py:55-296);
    #     matplotlib runs on the Agg backend, only the returned arrays are kept
    import ContourPlot as RC
    import matplotlib.pyplot as plt
    cp = {}
    verts = np.array([[0.0, -0.5], [0.0, -0.2], [0.0, 0.2], [0.0, 0.5], [2.0, 0.5], [2.0, -0.5]])
    obs = [np.array([[0.5, -0.2], [0.8, -0.2], [0.8, 0.1], [0.5, 0.1]])]
    f2 = lambda x, t=0.0: (np.sin(3 * x[:, 0:1]) * np.cos(2 * x[:, 1:2]) + t)
    f1 = lambda x, t: np.sin(np.pi * x) * np.exp(-t)
    for key, dom, tI in (('2dt', RD.PolygonDomain2D(verts), [0, 1.5]), ('2d', RD.PolygonDomain2D(verts), None),
                         ('2dobs', RD.PolygonDomain2D(verts, obs), [0, 1.5]), ('1dt', RD.Domain1D(), [0, 2.0])):
        c = RC.ContourPlot(dom, tI)
        cp[key + '_X'], cp[key + '_Y'], cp[key + '_out'], cp[key + '_he'] = c.X_coord, c.Y_coord, c.isOutside, c.he
        cp[key + '_xx'], cp[key + '_yy'] = c.xx, c.yy
        if key == '1dt':
            cp[key + '_field'] = c.conPlot(f1)
            c.snap1Dt(f1, 0.7)
            cp[key + '_snap'] = f1(c.x_coord, 0.7)
        elif key == '2d':
            cp[key + '_field'] = c.conPlot(f2)
        else:
            cp[key + '_field'] = c.conPlot(f2, 0.4, fill_val=-7.0)
        plt.close('all')
    np.savez_compressed(os.path.join(OUT, 'contour.npz'), **cp)
    print('golden fixtures written to', OUT)


if __name__ == '__main__':
    main()
