"""
activationFun='tanh' (a documented option of the kept constructor, /root/reference/VarNet.py:97; Keras
`Dense(..., activation='tanh')`, TFModel.py:208-221) on every kernel path, against the oracle with torch.tanh.
"""
import numpy as np
import pytest
import torch

from oracle import tf1_graph as og
from tests.test_engine_gpu import synth, LOSS_RTOL, GRAD_RTOL, LVEC_RTOL, CASES

pytestmark = pytest.mark.gpu


def engine(d_in, dim, widths, q, source, integW, kernel):
    from varnet_amd.engine import VNEngine
    return VNEngine(dim, d_in, widths, True, q, isSource=source, integWflag=integW, kernel=kernel,
                    activationFun='tanh')


TCASES = [CASES[i] for i in (0, 1, 2, 4, 5, 7, 9, 11, 12, 14, 15, 18, 19, 20, 22, 23, 24, 25)]


@pytest.mark.parametrize('kernel', [1, 0], ids=['generic', 'auto'])
@pytest.mark.parametrize('case', TCASES)
def test_tanh_loss_and_grad_parity(case, kernel):
    d_in, dim, widths, q, n_k, nB, bDof, source, integW, detJvec = case
    d = synth(2, d_in, dim, widths, q, n_k, nB, bDof, source, integW, detJvec)
    eng = engine(d_in, dim, widths, q, source, integW, kernel)
    if kernel == 0 and len(widths) <= (5 if max(widths) > 50 else 6):
        assert eng.kernel_path()[0] == 3                       # the 8-wave fused kernel carries tanh
    eng.init_params(seed=3)
    flat = eng.get_params() + 0.05 * np.random.default_rng(5).standard_normal(eng.P).astype(np.float32)
    eng.set_params(flat)
    eng.set_fe_table(d['N1'], d['dNt1'], d['integW'])
    eng.set_interior(0, d['Input'], d['gcoef'], d['source'], n_k=n_k, detJ=d['detJ'])
    eng.set_bic(d['biInput'], d['biLabel'], bDof, 2.0)
    eng.set_weights(d['w'])
    f64 = lambda a: None if a is None else a.astype(np.float64)
    ref, gref = og.loss_and_grad(
        flat.astype(np.float64), d_in, widths, torch.float64, Input=f64(d['Input']), gcoef=f64(d['gcoef']),
        source=f64(d['source']), N=f64(d['N']), dNt=f64(d['dNt']), integW=f64(d['integW']), intShape=[n_k, q],
        detJ=(f64(d['detJ']) if detJvec else float(d['detJ'])), detJvec=detJvec, biInput=f64(d['biInput']),
        biLabel=f64(d['biLabel']), bDof=bDof, biDimVal=2.0, w=d['w'], dim=dim, time_dependent=True,
        is_source=source, integWflag=integW, activation='tanh')
    out, lv = eng.eval_loss(0, lossVec=True)
    for got, key in zip(out, ['loss', 'BCloss', 'ICloss', 'varLoss']):
        assert abs(got - ref[key]) <= LOSS_RTOL * abs(ref[key]) + 1e-7, (key, got, ref[key])
    lref = ref['lossVec'].reshape(-1)
    assert np.max(np.abs(lv.cpu().numpy() - lref)) <= LVEC_RTOL * np.max(np.abs(lref))
    gb = eng.bind_grad_buffer()
    eng.grad(0)
    torch.cuda.synchronize()
    g = gb.cpu().numpy()
    assert abs(g[eng.P] - ref['loss']) <= LOSS_RTOL * abs(ref['loss'])
    assert np.max(np.abs(g[:eng.P] - gref)) / np.max(np.abs(gref)) <= GRAD_RTOL
    eng.close()


def test_tanh_forward_and_residual_parity():
    d_in, dim, widths = 3, 2, [10, 20, 30]
    rng = np.random.default_rng(0)
    n = 1000
    X = rng.uniform(-1, 1, (n, d_in))
    diff = rng.uniform(0.1, 1, (n, 1)); vel = rng.standard_normal((n, dim))
    src = rng.standard_normal((n, 1)); ddx = rng.standard_normal((n, dim))
    eng = engine(d_in, dim, widths, 64, False, False, 0)
    eng.init_params(seed=11)
    flat = eng.get_params().astype(np.float64)
    uref, rref = og.residual(flat, d_in, widths, torch.float64, X, diff, vel, src, ddx, dim, True, activation='tanh')
    u32 = eng.forward(X.astype(np.float32)).cpu().numpy()
    assert np.max(np.abs(u32 - uref[:, 0])) < 2e-6 * max(1, np.max(np.abs(uref)))
    u64 = eng.forward_f64(X).cpu().numpy()
    assert np.max(np.abs(u64 - uref[:, 0])) < 1e-13
    u, r = eng.residual(X, diff, vel, src, ddx, fp64=True)
    assert np.max(np.abs(r.cpu().numpy() - rref[:, 0])) < 1e-11 * max(1, np.max(np.abs(rref)))
    u, r = eng.residual(X.astype(np.float32), diff, vel, src, ddx, fp64=False)
    assert np.max(np.abs(r.cpu().numpy() - rref[:, 0])) < 5e-5 * max(1, np.max(np.abs(rref)))
    eng.close()


def test_tanh_training_through_varnet(tmp_path):
    """Constructor option end to end: VarNet(..., activationFun='tanh').train on the Operator_1Dt problem."""
    from tests.test_varnet_host import cExact, pi
    from varnet_amd import ADPDE, Domain1D, VarNet
    pde = ADPDE(Domain1D(), diff=0.1 / pi, vel=1.0, timeDependent=True, tInterval=[0, 2.0],
                IC=lambda x: -np.sin(pi * x), cEx=cExact)
    vn = VarNet(pde, layerWidth=[20, 20, 20], activationFun='tanh', discNum=20, bDiscNum=None, tDiscNum=60)
    res = vn.train(str(tmp_path), weight=[10., 10., 1.], epochNum=300, saveFreq=100, verbose=False)
    assert res.lossAll[-1] < 0.7 * res.lossAll[0]
    flat = vn.engine.get_params().astype(np.float64)
    uref = og.forward(flat, 2, [20, 20, 20], torch.float64, vn.fixData.uniform_input, activation='tanh')
    assert np.max(np.abs(vn.evaluate() - uref)) < 5e-6
    assert 'tanh' in open(str(tmp_path / 'caseData.txt')).read()
    with pytest.raises(ValueError):          # 'activation function list is incompatible with number of layers!' (TFModel.py:117)
        VarNet(pde, layerWidth=[8, 8], activationFun=['tanh', 'sigmoid', 'tanh'], discNum=5, bDiscNum=None, tDiscNum=6)
    vn.engine.close()
    # one entry per layer with different entries is legal in the reference: it trains on the layer-by-layer route
    vm = VarNet(pde, layerWidth=[8, 8], activationFun=['tanh', 'sigmoid'], discNum=5, bDiscNum=None, tDiscNum=6)
    rm = vm.train(str(tmp_path / 'mixed'), weight=[10., 10., 1.], epochNum=20, saveFreq=10, verbose=False)
    assert vm.engine.kernel_path()[0] == 4 and np.isfinite(rm.lossAll).all()
    vm.engine.close()
