"""
CPU tests: the build's own problem layer (FE tables, meshes, utilities, ADPDE, MOR) against
golden fixtures generated from the reference's importable NumPy modules
(oracle/gen_golden.py), plus known answers.
"""
import os
import numpy as np
import pytest

from varnet_amd.finite_element import FE
from varnet_amd.domain import Domain1D, PolygonDomain2D
from varnet_amd.adpde import ADPDE
from varnet_amd.mor import MOR
from varnet_amd.utility import UF

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
uf = UF()
TOL = dict(rtol=1e-13, atol=1e-14)


@pytest.mark.parametrize('D,ip', [(1, 2), (2, 2), (3, 2), (1, 3), (2, 3), (3, 3)])
def test_fe_tables_match_reference(D, ip):
    g = np.load(os.path.join(G, 'fe_tables.npz'))
    k = 'D%d_ip%d_' % (D, ip)
    f = FE(D, ip)
    assert np.array_equal(f.basMultiInd, g[k + 'basMultiInd'])
    np.testing.assert_allclose(f.IntegP, g[k + 'IntegP'], **TOL)
    np.testing.assert_allclose(f.basVal, g[k + 'basVal'], **TOL)
    np.testing.assert_allclose(f.basDeriVal, g[k + 'basDeriVal'], **TOL)
    np.testing.assert_allclose(f.elemCoord, g[k + 'elemCoord'], **TOL)
    np.testing.assert_allclose(f.delta, g[k + 'delta'], **TOL)
    if g[k + 'IntegW'].size == 0:
        assert f.IntegW is None
    else:
        np.testing.assert_allclose(f.IntegW, g[k + 'IntegW'], **TOL)
    np.testing.assert_allclose(f.massVec, g[k + 'massVec'], **TOL)          # FiniteElement.py:438-499
    np.testing.assert_allclose(f.massDelta, g[k + 'massDelta'], **TOL)
    np.testing.assert_allclose(np.sum(f.massVec), 2.0 ** D, rtol=1e-13)     # the hat function integrates to 2^D (reference cell)
    integNum, nT, detJ, delta, iw, N, dN = f.basisTot(3, g[k + 'bt_hVec'])
    np.testing.assert_allclose([integNum, nT, detJ], g[k + 'bt_scalars'], **TOL)
    np.testing.assert_allclose(delta, g[k + 'bt_delta'], **TOL)
    np.testing.assert_allclose(N, g[k + 'bt_N'], **TOL)
    np.testing.assert_allclose(dN, g[k + 'bt_dN'], **TOL)
    if g[k + 'bt_intWeight'].size == 0:
        assert iw is None
    else:
        np.testing.assert_allclose(iw, g[k + 'bt_intWeight'], **TOL)


@pytest.mark.parametrize('D,ip', [(1, 2), (2, 2), (3, 2), (2, 3), (3, 3)])
def test_fe_known_answers(D, ip):
    """Partition of unity (FiniteElement.py:547-551) and hat-function integral = prod(h)."""
    f = FE(D, ip)
    np.testing.assert_allclose(f.basVal.sum(axis=0), 1.0, atol=1e-14)
    np.testing.assert_allclose(f.basDeriVal.sum(axis=1), 0.0, atol=1e-14)
    h = np.array([0.3, 0.05, 0.7])[:D]
    integNum, detJ, delta, iw, N, dN = f.basisTable(h)
    w = np.ones(integNum) if iw is None else iw.reshape(-1)
    np.testing.assert_allclose(detJ * np.sum(w * N), np.prod(h), rtol=1e-13)
    # quadrature points stay inside the support [-1,1]*h
    assert np.all(np.abs(delta) < 1.0)


def test_meshes_match_reference():
    g = np.load(os.path.join(G, 'meshes.npz'))
    m = Domain1D().getMesh(20)
    np.testing.assert_allclose(m.coordinates, g['d1_coord'], **TOL)
    np.testing.assert_allclose(m.he, g['d1_he'], **TOL)
    assert np.array_equal(m.bdof, g['d1_bdof'])
    np.testing.assert_allclose(m.bCoordinates, g['d1_bcoord'], **TOL)
    m = Domain1D(np.array([0.5, 3.0])).getMesh(7)
    np.testing.assert_allclose(m.coordinates, g['d1b_coord'], **TOL)
    np.testing.assert_allclose(m.he, g['d1b_he'], **TOL)

    dom = PolygonDomain2D(g['p2_vertices'])
    m = dom.getMesh([8, 4], 4)
    np.testing.assert_allclose(m.coordinates, g['p2_coord'], **TOL)
    np.testing.assert_allclose(m.he, g['p2_he'], **TOL)
    assert np.array_equal(np.array(m.bdof), g['p2_bdof'])
    np.testing.assert_allclose(np.vstack(m.bCoordinates), g['p2_bcoord'], **TOL)
    np.testing.assert_allclose(dom.measure, g['p2_measure'], **TOL)
    np.testing.assert_allclose(dom.lim, g['p2_lim'], **TOL)
    np.testing.assert_allclose(dom.boundryGeom, g['p2_bgeom'], **TOL)
    np.testing.assert_allclose(dom.scaleCoord(g['p2o_pts']), g['p2_scale'], **TOL)

    dom = PolygonDomain2D(g['p2_vertices'], [g['p2o_obs']])
    m = dom.getMesh([16, 9], 3)
    np.testing.assert_allclose(m.coordinates, g['p2o_coord'], **TOL)
    assert np.array_equal(np.array(m.bdof), g['p2o_bdof'])
    np.testing.assert_allclose(np.vstack(m.bCoordinates), g['p2o_bcoord'], **TOL)
    assert m.bIndNum == int(g['p2o_bIndNum'])
    assert np.array_equal(dom.isInside(g['p2o_pts']), g['p2o_inside'])


def test_utilities_match_reference():
    g = np.load(os.path.join(G, 'utility.npz'))
    np.testing.assert_array_equal(uf.pairMats(g['pm_a'], g['pm_b']), g['pm_ab'])
    np.testing.assert_array_equal(uf.pairMats(g['pm_a'], g['pm_b'], reverse=True), g['pm_ab_rev'])
    np.testing.assert_allclose(uf.l2Err(g['l2_x'], g['l2_y']), g['l2'], **TOL)
    np.testing.assert_allclose(uf.polyArea(g['poly']), g['polyArea'], **TOL)
    assert uf.isempty([]) and uf.isempty({}) and not uf.isempty([0])
    assert uf.isnone(None) and uf.isnone([1, None]) and not uf.isnone([]) and not uf.isnone(0.0)
    assert uf.vstack([[], np.ones((1, 2))]).shape == (1, 2) and uf.vstack([[], []]) == []


def test_adpde_matches_reference():
    g = np.load(os.path.join(G, 'adpde.npz'))
    verts = np.load(os.path.join(G, 'meshes.npz'))['p2_vertices']
    BC = [[], [0.0, 1.0, 1.0], [], [1.0, 0.0, 2.0], [1.0, 2.0, 0.5], []]
    pde = ADPDE(PolygonDomain2D(verts), diff=1e-3, vel=[1., 0.], tInterval=[0, 1.5], BCs=BC, IC=0.0)
    xs, ts = g['x'], g['t']
    assert list(g['BCtype']) == pde.BCtype
    np.testing.assert_allclose(np.array([[b[0], b[1]] for b in pde.BCs], dtype=float), g['BCab'])
    np.testing.assert_allclose(np.hstack([b[2](xs, ts) for b in pde.BCs]), g['BCg'])
    np.testing.assert_allclose(pde.diffFun(xs, ts), g['diff'])
    np.testing.assert_allclose(pde.velFun(xs, ts), g['vel'])
    np.testing.assert_allclose(pde.sourceFun(xs, ts), g['source'])
    np.testing.assert_allclose(pde.d_diffFun(xs, ts), g['d_diff'])
    np.testing.assert_allclose(pde.IC(xs), g['IC'])
    assert pde.timeDependent == bool(g['timeDependent'])
    with pytest.raises(ValueError):
        ADPDE(PolygonDomain2D(verts), diff=1e-3, vel=[1., 0.], tInterval=[0, 1.5], BCs=BC)   # IC missing
    with pytest.raises(ValueError):
        ADPDE(PolygonDomain2D(verts), diff='x', vel=[1., 0.])


def test_mor_matches_reference():
    g = np.load(os.path.join(G, 'mor.npz'))

    def diffFun(x, t=0, D=0.01):
        return D * np.ones([np.shape(x)[0], 1])

    def discDiff(discNum=6):
        return np.array([0.003 * (11 ** (n / (discNum - 1))) for n in range(discNum)])[np.newaxis].T

    mor = MOR(diffFun, ['D'], [[0.003, 0.033]])
    da = mor.discretizeArg(discDiff)
    np.testing.assert_allclose(da[0], g['m1_disc0'], **TOL)
    np.testing.assert_array_equal(mor.argIndex(da), g['m1_argInd'])
    assert list(g['m1_varNum']) == mor.varNum

    def velFun(x, t=0, a=1.0, b=2.0):
        return a * np.ones([np.shape(x)[0], 1])

    mor2 = MOR([diffFun, velFun], [['D'], ['b', 'a']], [[[0.1, 0.2]], [[3.0, 4.0], [1.0, 2.0]]])
    da2 = mor2.discretizeArg([3, [2, 4]])
    np.testing.assert_allclose(da2[0], g['m2_disc0'], **TOL)
    np.testing.assert_allclose(da2[1], g['m2_disc1'], **TOL)
    np.testing.assert_array_equal(mor2.argIndex(da2), g['m2_argInd'])
    assert list(g['m2_names1']) == mor2.ArgNames[1]
    with pytest.raises(ValueError):
        MOR(diffFun, ['nope'], [[0, 1]])


def test_contour_grid_and_fields_match_reference():
    """varnet_amd.ContourPlot reproduces the reference's plotting grid and the arrays conPlot / snap1Dt draw
    (ContourPlot.py:55-296; fixture from the reference's own module, Agg backend)."""
    from varnet_amd.contour import ContourPlot
    GD = globals()['G']
    G = np.load(os.path.join(GD, 'contour.npz'))
    verts = np.array([[0.0, -0.5], [0.0, -0.2], [0.0, 0.2], [0.0, 0.5], [2.0, 0.5], [2.0, -0.5]])
    obs = [np.array([[0.5, -0.2], [0.8, -0.2], [0.8, 0.1], [0.5, 0.1]])]
    f2 = lambda x, t=0.0: (np.sin(3 * x[:, 0:1]) * np.cos(2 * x[:, 1:2]) + t)
    f1 = lambda x, t: np.sin(np.pi * x) * np.exp(-t)
    for key, dom, tI in (('2dt', PolygonDomain2D(verts), [0, 1.5]), ('2d', PolygonDomain2D(verts), None),
                         ('2dobs', PolygonDomain2D(verts, obs), [0, 1.5]), ('1dt', Domain1D(), [0, 2.0])):
        c = ContourPlot(dom, tI)
        np.testing.assert_allclose(c.X_coord, G[key + '_X'], rtol=0, atol=1e-15)
        np.testing.assert_allclose(c.Y_coord, G[key + '_Y'], rtol=0, atol=1e-15)
        np.testing.assert_array_equal(c.isOutside, G[key + '_out'])
        np.testing.assert_allclose(c.he, G[key + '_he'], rtol=1e-15)
        np.testing.assert_allclose(c.xx, G[key + '_xx'], rtol=0, atol=1e-15)
        np.testing.assert_allclose(c.yy, G[key + '_yy'], rtol=0, atol=1e-15)
        if key == '1dt':
            np.testing.assert_allclose(c.field(f1), G[key + '_field'], rtol=1e-14, atol=1e-15)
            np.testing.assert_allclose(c.snap(f1, 0.7)[1], G[key + '_snap'], rtol=1e-14, atol=1e-15)
        elif key == '2d':
            np.testing.assert_allclose(c.field(f2), G[key + '_field'], rtol=1e-14, atol=1e-15)
        else:
            np.testing.assert_allclose(c.field(f2, 0.4, fill_val=-7.0), G[key + '_field'], rtol=1e-14, atol=1e-15)
    assert G['2dobs_out'].sum() > G['2dt_out'].sum()          # the obstacle really masks grid points


def test_rejection_sampling_matches_reference():
    """uf.rejectionSampling consumes the NumPy stream like the reference's (UtilityFunc.py:342-404): same seed, same
    accepted samples -- one segment, and two segments with dofT."""
    Gr = np.load(os.path.join(G, 'rejection.npz'))
    uf = UF()
    grid = [np.linspace(-1, 1, 41).reshape(-1, 1)]

    def dens(x=None):
        if x is None:
            x = grid[0]
        return np.exp(-8.0 * (x[:, 0:1] - 0.3) ** 2) + 0.05

    np.random.seed(2024)
    np.testing.assert_array_equal(uf.rejectionSampling(dens, lambda: np.random.uniform(-1, 1, [50, 1]), 37), Gr['one'])
    grid[0] = np.vstack([np.linspace(-1, 1, 30).reshape(-1, 1), np.linspace(-1, 1, 20).reshape(-1, 1)])
    np.random.seed(7)
    np.testing.assert_array_equal(uf.rejectionSampling(dens, lambda: np.random.uniform(-1, 1, [50, 1]), [11, 9], [30, 20]), Gr['two'])
