"""
The one-wave-per-SIMD kernel (vn_fusedpw.hip: 4 waves x 512 registers, every wave contracts the weight gradient of its own
16 points, no publish / release barriers) against the fp64 oracle at the bars of tests/test_engine_gpu.py, and against the
8-wave kernel it replaces for these shapes.  Shapes it serves: hidden width 33..50, 1..5 hidden layers, d_in <= 3, integNum
dividing 64.
"""
import numpy as np
import pytest
import torch

from tests.test_engine_gpu import synth, oracle_eval, LOSS_RTOL, GRAD_RTOL, LVEC_RTOL

pytestmark = pytest.mark.gpu

CASES = [
    # d_in dim widths                 integNum n_k   nB   bDof source integW detJvec act
    (3, 2, [50, 50, 50, 50, 50],      64,      9,    77,  40,  False, False, False, 'sigmoid'),   # BASELINE config 3's net
    (2, 1, [50, 50, 50, 50],          16,      41,   50,  30,  False, False, False, 'sigmoid'),   # config 2's net: R_k inside a wave
    (3, 2, [50, 50, 50, 50, 50],      64,      700,  300, 140, True,  False, True,  'sigmoid'),   # several tiles per workgroup, source, detJ per test function
    (3, 2, [40, 33],                  64,      5,    33,  20,  True,  True,  False, 'sigmoid'),   # ragged widths, integW
    (2, 1, [50],                      16,      21,   19,  7,   False, False, False, 'sigmoid'),   # one hidden layer
    (3, 1, [45, 50, 38],              32,      11,   64,  64,  False, False, False, 'tanh'),      # two waves per test function, tanh, no IC rows
    (1, 1, [50, 50],                  4,       33,   10,  0,   False, True,  False, 'sigmoid'),   # d_in 1, no BC rows
    (3, 2, [50, 50, 50, 50, 50],      1,       130,  70,  33,  False, False, False, 'sigmoid'),   # integNum 1
]


def _engine(case, pw, monkeypatch):
    d_in, dim, widths, integNum, n_k, nB, bDof, source, integW, detJvec, act = case
    d = synth(7, d_in, dim, widths, integNum, n_k, nB, bDof, source, integW, detJvec)
    from varnet_amd.engine import VNEngine
    monkeypatch.setenv('VN_PW', '1' if pw else '0')
    eng = VNEngine(dim, d_in, widths, True, integNum, isSource=source, integWflag=integW, activationFun=act)
    eng.init_params(seed=3)
    flat = eng.get_params()
    flat = flat + 0.05 * np.random.default_rng(5).standard_normal(flat.size).astype(np.float32)
    eng.set_params(flat)
    eng.set_fe_table(d['N1'], d['dNt1'], d['integW'])
    eng.set_interior(0, d['Input'], d['gcoef'], d['source'], n_k=n_k, detJ=d['detJ'])
    eng.set_bic(d['biInput'], d['biLabel'], bDof, 2.0)
    eng.set_weights(d['w'])
    return eng, d, flat


@pytest.mark.parametrize('case', CASES)
def test_loss_and_grad_parity_pw(case, monkeypatch):
    d_in, dim, widths, integNum, n_k, nB, bDof, source, integW, detJvec, act = case
    eng, d, flat = _engine(case, True, monkeypatch)
    eng.profile_begin()
    gb = eng.bind_grad_buffer()
    eng.grad(0)
    torch.cuda.synchronize()
    _, _, name = eng.profile_end()
    assert name.startswith('vn_fusedpw_kernel<%d' % len(widths)), name              # this kernel ran, not the 8-wave one
    g = gb.cpu().numpy().astype(np.float64)
    import oracle.tf1_graph as og
    kw = dict(Input=d['Input'].astype(np.float64), gcoef=d['gcoef'].astype(np.float64),
              source=None if d['source'] is None else d['source'].astype(np.float64),
              N=d['N'].astype(np.float64), dNt=d['dNt'].astype(np.float64),
              integW=None if d['integW'] is None else d['integW'].astype(np.float64), intShape=[n_k, integNum],
              detJ=(d['detJ'].astype(np.float64) if detJvec else float(d['detJ'])), detJvec=detJvec,
              biInput=d['biInput'].astype(np.float64), biLabel=d['biLabel'].astype(np.float64), bDof=bDof, biDimVal=2.0,
              w=d['w'], dim=dim, time_dependent=True, is_source=source, integWflag=integW)
    ref, gref = og.loss_and_grad(flat.astype(np.float64), d_in, widths, torch.float64, activation=act, **kw) \
        if act != 'sigmoid' else oracle_eval(flat, d, d_in, dim, widths, integNum, n_k, bDof, source, integW, detJvec)
    P = eng.P
    assert abs(g[P] - ref['loss']) <= LOSS_RTOL * abs(ref['loss']), (g[P], ref['loss'])
    for i, key in ((1, 'BCloss'), (2, 'ICloss'), (3, 'varLoss')):
        assert abs(g[P + i] - ref[key]) <= LOSS_RTOL * abs(ref[key]) + 1e-7, (key, g[P + i], ref[key])
    err = np.max(np.abs(g[:P] - gref)) / np.max(np.abs(gref))
    assert err <= GRAD_RTOL, err
    eng.grad(0)                                                                       # run-to-run reproducible
    torch.cuda.synchronize()
    assert np.array_equal(gb.cpu().numpy().astype(np.float64), g)
    # the 8-wave kernel on the same inputs: two fp32 programs with different summation orders
    eng8, _, _ = _engine(case, False, monkeypatch)
    gb8 = eng8.bind_grad_buffer()
    eng8.grad(0)
    torch.cuda.synchronize()
    g8 = gb8.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(g8[:P] - g[:P])) <= 3e-5 * np.max(np.abs(g[:P])) and abs(g8[P] - g[P]) <= 1e-5 * abs(g[P])
    eng.close()
    eng8.close()


def test_training_trajectory_pw_matches_the_8_wave_kernel(monkeypatch):
    """200 TF-1 Adam steps from the same start on both kernels (train_epoch path: gradient + folded optimizer)."""
    case = (3, 2, [50, 50, 50, 50, 50], 64, 120, 90, 50, False, False, False, 'sigmoid')
    out = []
    for pw in (True, False):
        eng, d, flat = _engine(case, pw, monkeypatch)
        losses = torch.zeros(200, device='cuda')
        for i in range(200):
            eng.train_step(0, losses[i:i + 1])
        torch.cuda.synchronize()
        out.append((losses.cpu().numpy().astype(np.float64), eng.get_params().astype(np.float64)))
        eng.close()
    (la, ta), (lb, tb) = out
    assert np.max(np.abs(la - lb) / np.abs(lb)) <= 1e-3
    assert np.max(np.abs(ta - tb)) <= 1e-3 * np.max(np.abs(tb))
    assert la[-1] < la[0]
