"""
CPU tests that pin the oracle itself (it cannot be pinned on the TF-1.10 binary, see
oracle/tf1_graph.py header): finite differences of every derivative, agreement between the
autograd restatement of the reference graph and the hand-derived tangent formulation the HIP
kernels implement, TF-1 Adam known answer, glorot limits, shard rule.
"""
import numpy as np
import torch

from oracle import tf1_graph as og
from oracle import tangent_ref as tr


def _case(seed=0, d_in=3, dim=2, widths=(7, 5, 6), q=8, n_k=5, nB=11, bDof=7):
    rng = np.random.default_rng(seed)
    n = n_k * q
    flat = og.glorot_init(d_in, list(widths), 1).astype(np.float64) + 0.1 * rng.standard_normal(og.param_count(d_in, list(widths)))
    kw = dict(Input=rng.uniform(-1, 1, (n, d_in)), gcoef=rng.standard_normal((n, dim)),
              source=rng.standard_normal((n, 1)), N=rng.uniform(0, 1, (n, 1)), dNt=rng.standard_normal((n, 1)),
              integW=rng.uniform(.5, 1, (1, q)), intShape=[n_k, q], detJ=0.37, detJvec=False,
              biInput=rng.uniform(-1, 1, (nB, d_in)), biLabel=rng.standard_normal((nB, 1)), bDof=bDof,
              biDimVal=2.0, w=np.array([3., 2., 5.]), dim=dim, time_dependent=True, is_source=True,
              integWflag=True)
    return flat, list(widths), d_in, dim, q, n_k, kw


def test_autograd_graph_equals_tangent_formulation():
    flat, widths, d_in, dim, q, n_k, kw = _case()
    res, g = og.loss_and_grad(flat, d_in, widths, torch.float64, **kw)
    res2, g2 = tr.loss_and_grad(flat, d_in, widths, dim, kw['Input'], kw['gcoef'], kw['source'], kw['N'],
                                kw['dNt'], kw['integW'], n_k, q, kw['detJ'], kw['biInput'], kw['biLabel'],
                                kw['bDof'], kw['biDimVal'], kw['w'])
    for k in ('loss', 'BCloss', 'ICloss', 'varLoss'):
        np.testing.assert_allclose(res2[k], res[k], rtol=1e-13)
    np.testing.assert_allclose(res2['lossVec'], res['lossVec'].reshape(-1), rtol=1e-12)
    np.testing.assert_allclose(g2, g, rtol=1e-10, atol=1e-12)


def test_parameter_gradient_finite_difference():
    flat, widths, d_in, dim, q, n_k, kw = _case(seed=3)
    _, g = og.loss_and_grad(flat, d_in, widths, torch.float64, **kw)
    rng = np.random.default_rng(0)
    for i in rng.choice(flat.size, 12, replace=False):
        e = 1e-6
        fp, fm = flat.copy(), flat.copy()
        fp[i] += e
        fm[i] -= e
        lp = og.loss_and_grad(fp, d_in, widths, torch.float64, **kw)[0]['loss']
        lm = og.loss_and_grad(fm, d_in, widths, torch.float64, **kw)[0]['loss']
        np.testing.assert_allclose((lp - lm) / (2 * e), g[i], rtol=2e-6, atol=1e-7)


def test_input_gradient_and_laplacian_finite_difference():
    d_in, dim, widths = 3, 2, [6, 7]
    rng = np.random.default_rng(1)
    flat = og.glorot_init(d_in, widths, 4).astype(np.float64)
    X = rng.uniform(-1, 1, (5, d_in))
    diff = rng.uniform(.1, 1, (5, 1)); vel = rng.standard_normal((5, dim))
    src = rng.standard_normal((5, 1)); ddx = rng.standard_normal((5, dim))
    u, res = og.residual(flat, d_in, widths, torch.float64, X, diff, vel, src, ddx, dim, True)
    f = lambda Z: og.forward(flat, d_in, widths, torch.float64, Z)
    h = 1e-4
    grad = np.zeros((5, d_in)); lap = np.zeros((5, 1))
    for d in range(d_in):
        e = np.zeros(d_in); e[d] = h
        grad[:, d:d + 1] = (f(X + e) - f(X - e)) / (2 * h)
        if d < dim:
            lap += (f(X + e) - 2 * f(X) + f(X - e)) / h ** 2
    ref = -grad[:, dim:dim + 1] + diff * lap - ((vel - ddx) * grad[:, :dim]).sum(1, keepdims=True) + src
    np.testing.assert_allclose(u, f(X), rtol=1e-14)
    np.testing.assert_allclose(res, ref, rtol=1e-5, atol=1e-6)


def test_detjvec_and_no_source_branches():
    flat, widths, d_in, dim, q, n_k, kw = _case(seed=5)
    kw['detJ'] = np.random.default_rng(2).uniform(.1, .2, (n_k, 1))
    kw['detJvec'] = True
    kw['is_source'] = False
    kw['integWflag'] = False
    res, g = og.loss_and_grad(flat, d_in, widths, torch.float64, **kw)
    res2, g2 = tr.loss_and_grad(flat, d_in, widths, dim, kw['Input'], kw['gcoef'], None, kw['N'], kw['dNt'],
                                None, n_k, q, kw['detJ'], kw['biInput'], kw['biLabel'], kw['bDof'],
                                kw['biDimVal'], kw['w'])
    np.testing.assert_allclose(res2['loss'], res['loss'], rtol=1e-13)
    np.testing.assert_allclose(g2, g, rtol=1e-10, atol=1e-12)
    # detJ multiplies R^2 once (TFModel.py:571-577): lossVec = detJ * R^2
    np.testing.assert_allclose(res['varLoss'], res['lossVec'].sum(), rtol=1e-13)


def test_tf1_adam_known_answer():
    """First step of TF-1 Adam moves every coordinate by lr*g/(|g|+eps*sqrt(1-b2)) ~ lr*sign(g)."""
    g = np.array([0.5, -2.0, 1e-3])
    ad = og.TF1Adam(3, lr=1e-3, dtype=np.float64)
    th = ad.step(np.zeros(3), g)
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    exp = -lr_t * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
    np.testing.assert_allclose(th, exp, rtol=1e-14)
    np.testing.assert_allclose(np.abs(th), 1e-3, rtol=1e-3)
    th2 = ad.step(th, g)
    assert ad.t == 2 and np.all(np.abs(th2) > np.abs(th))


def test_glorot_limits_and_determinism():
    d_in, widths = 3, [50, 50]
    a = og.glorot_init(d_in, widths, 7)
    b = og.glorot_init(d_in, widths, 7)
    c = og.glorot_init(d_in, widths, 8)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    W1 = a[:150]
    lim = np.sqrt(6 / 53)
    assert np.all(np.abs(W1) <= lim) and np.abs(W1).max() > 0.8 * lim
    assert np.all(a[150:200] == 0)                        # biases zero
    assert a.size == og.param_count(d_in, widths) == 150 + 50 + 2500 + 50 + 50 + 1


def test_shard_rule():
    assert [og.shard_rows(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert og.shard_rows(100000, 7, 8) == (87500, 100000)


def test_tf1_rmsprop_known_answer():
    """First steps of TF-1 RMSProp by hand: ms starts at ones, ms1 = 1 + (g^2-1)*0.1, step = lr*g/sqrt(ms1+1e-10)."""
    opt = og.TF1RMSProp(2, lr=1e-2, dtype=np.float64)
    th = np.array([1.0, -2.0])
    g = np.array([3.0, 0.5])
    th1 = opt.step(th, g)
    ms1 = 1.0 + (g * g - 1.0) * 0.1
    np.testing.assert_allclose(opt.ms, ms1, rtol=1e-15)
    np.testing.assert_allclose(th1, th - 1e-2 * g / np.sqrt(ms1 + 1e-10), rtol=1e-15)
    th2 = opt.step(th1, g)
    ms2 = ms1 + (g * g - ms1) * 0.1
    np.testing.assert_allclose(th2, th1 - 1e-2 * g / np.sqrt(ms2 + 1e-10), rtol=1e-15)
