"""
The only reference-held numbers on the device path: the tabulated exact values (Mojtabi & Deville) that
/root/reference/Operator_1Dt.py:113-128 and /root/reference/Operator_1DtMOR.py:117-150 keep for kappa = 0.01/pi and
kappa = 0.005 (25 points each), and the reference's own acceptance metric l2Err(cEx, cApp) (Operator_1Dt.py:177-186,
Operator_1DtMOR.py:214-224).  tests/golden/exact_tables.npz holds the arrays (oracle/gen_golden_tables.py).

What is asserted on the GPU:
  1. the HIP engine and the fp32 oracle (the CPU restatement of TFModel.py:515-714 + TF-1 Adam), trained for the same
     fixed step budget from the same theta_0 on the Operator_1Dt set-up at kappa = 0.01/pi and on the Operator_1DtMOR
     set-up, give the same field at the 25 tabulated points (FIELD_BAR), i.e. the same distance from the table;
  2. a longer run of the HIP engine alone brings the trained field within a stated distance of the table (the
     reference publishes no value of its own metric, so the distances are the ones measured here: TABLE_BAR_*).
Pairing: kappa = 0.01/pi <-> cExD3, kappa = 0.005 <-> cExD4; the reference's script swaps them
(Operator_1DtMOR.py:216-224) -- not copied.
The measured numbers go to gpurun_out/r3_exact_tables.json (copied to profiles/ by hand).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import tf1_graph as og
from varnet_amd import ADPDE, Domain1D, MOR, VarNet
from varnet_amd.utility import UF

uf = UF()
pi = np.pi
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'exact_tables.npz')

# Measured (one MI355X, profiles/r3_exact_tables.json): field difference 3.1e-6 (Operator_1Dt, 3000 steps) and 8.1e-7
# (Operator_1DtMOR, 240 steps); l2Err against the table 0.64599 on BOTH sides after 3000 steps, 0.326 after the 120 000-epoch
# run of the script's own settings (0.159 on the points x <= 0.9, i.e. outside the boundary layer at x = 1 that a 20-unit
# net on a 20 x 300 grid resolves last); Operator_1DtMOR 0.58 / 0.56 after 1500 of its epochs (loss 9.4e5 -> 1.6e5; the
# reference lets it run to 500 000), longer runs in profiles/r3_exact_tables_long.txt.
FIELD_BAR = 1e-4            # max |u_hip - u_oracle| at the tabulated points after the fixed budget (values are O(1))
TABLE_BAR_1DT = 0.40        # l2Err(table, HIP field), Operator_1Dt set-up, 120 000 epochs
TABLE_BAR_MOR = 0.65        # l2Err(table, HIP field), Operator_1DtMOR set-up, 1500 epochs, both diffusivities


def tables():
    d = np.load(GOLD)
    return d['inpEx'], d['cEx_kappa_0p01_over_pi'], d['cEx_kappa_0p005'], d['kappa']


def record(key, value):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, 'r3_exact_tables.json')
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[key] = value
        json.dump(data, open(path, 'w'), indent=1, sort_keys=True)
    except OSError:
        pass


def test_exact_tables_fixture():
    """The fixture is the reference's data: 7 + 11 + 7 points at t = 0.8, 1.0, 1.6; both tables vanish on the
    outflow boundary x = 1 (homogeneous Dirichlet) and are close to each other away from the boundary layer."""
    inp, d3, d4, kappa = tables()
    assert inp.shape == (25, 2) and d3.shape == (25, 1) and d4.shape == (25, 1)
    np.testing.assert_array_equal(inp[:, 1], [0.8] * 7 + [1.0] * 11 + [1.6] * 7)
    np.testing.assert_array_equal(inp[:7, 0], [0.9, 0.94, 0.96, 0.98, 0.99, 0.999, 1.0])
    np.testing.assert_array_equal(inp[18:, 0], inp[:7, 0])
    np.testing.assert_allclose(kappa, [0.01 / pi, 0.005])
    assert d3[0, 0] == -0.30516 and d4[0, 0] == -0.29706 and d3[8, 0] == 0.98441 and d4[8, 0] == 0.95185
    on_boundary = inp[:, 0] == 1.0
    assert on_boundary.sum() == 3 and np.all(d3[on_boundary] == 0) and np.all(d4[on_boundary] == 0)
    inner = inp[:, 0] <= 0.9
    assert np.max(np.abs(d3[inner] - d4[inner])) < 0.05           # less diffusion, slightly larger amplitude
    assert np.all(np.abs(d3[inner]) >= np.abs(d4[inner]) - 1e-5)


# ---------------------------------------------------------------------------------------------------------
def _silence(vn):
    class Log:
        verbose = False

        def writeCase(self, s):
            pass
    vn.trainRes = Log()


def _op1dt_advective():
    """Operator_1Dt.py:144-161 with D = 0.01/pi (the case its table is for; cExact is disabled there, :137)."""
    pde = ADPDE(Domain1D(), diff=0.01 / pi, vel=1.0, timeDependent=True, tInterval=[0, 2.0],
                IC=lambda x: -np.sin(pi * x))
    return VarNet(pde, layerWidth=[20], discNum=20, bDiscNum=None, tDiscNum=300)


@pytest.mark.gpu
def test_operator_1dt_tables_hip_vs_oracle():
    from tests.test_operator_parity_gpu import run_both
    inp, d3, _, _ = tables()
    vn = _op1dt_advective()
    assert vn.fixData.nT == 96000 and vn.engine.P == 81
    steps = 3000
    gl, cl, th_g, th_c, _ = run_both(vn, [10., 10., 1.], steps)
    dev = np.abs(gl - cl) / np.abs(cl)
    u_g = vn.evaluate(x=inp[:, 0:1], t=inp[:, 1:2])                                  # Operator_1Dt.py:181
    u_c = og.forward(th_c.astype(np.float64), 2, [20], torch.float64, inp)
    e_g, e_c = uf.l2Err(d3, u_g), uf.l2Err(d3, u_c)
    fdiff = float(np.max(np.abs(u_g - u_c)))
    record('operator_1dt_kappa_0.01_over_pi_budget', dict(
        steps=steps, max_rel_loss_dev=float(dev.max()), l2Err_table_hip=float(e_g), l2Err_table_oracle=float(e_c),
        max_field_diff_at_table_points=fdiff, loss_first_last_hip=[float(gl[0]), float(gl[-1])],
        loss_first_last_oracle=[float(cl[0]), float(cl[-1])]))
    print('1Dt tables, %d steps: l2Err hip %.5f oracle %.5f, field diff %.2e, loss dev %.2e' % (steps, e_g, e_c, fdiff, dev.max()))
    assert dev.max() <= 1e-2
    assert fdiff <= FIELD_BAR
    assert abs(e_g - e_c) <= 1e-4
    vn.engine.close()


@pytest.mark.gpu
def test_operator_1dt_tables_long_run(tmp_path):
    """Operator_1Dt.py:170 as the script runs it (residual-driven sampling, adjustWeight), bounded to 120 000 epochs,
    then its own metric against its own table (Operator_1Dt.py:181-186)."""
    inp, d3, _, _ = tables()
    vn = _op1dt_advective()
    np.random.seed(0)
    res = vn.train(str(tmp_path), weight=[10., 10., 1.], smpScheme='optimal', adjustWeight=True, epochNum=120000,
                   saveFreq=2000, verbose=False)
    u = vn.evaluate(x=inp[:, 0:1], t=inp[:, 1:2])
    e = float(uf.l2Err(d3, u))
    interior = inp[:, 0] <= 0.9
    e_in = float(uf.l2Err(d3[interior], u[interior]))
    record('operator_1dt_kappa_0.01_over_pi_long', dict(epochs=len(res.lossAll), l2Err_table_hip=e,
                                                         l2Err_table_hip_x_le_0p9=e_in,
                                                         loss_first_last=[float(res.lossAll[0]), float(res.lossAll[-1])]))
    print('1Dt tables, long run: l2Err %.5f (x <= 0.9: %.5f) after %d epochs' % (e, e_in, len(res.lossAll)))
    assert e <= TABLE_BAR_1DT
    vn.engine.close()


# ---------------------------------------------------------------------------------------------------------
def _mor_setup():
    """Operator_1DtMOR.py:163-196."""
    def diffFun(x, t=0, D=0.01):
        return D * np.ones([np.shape(x)[0], 1])

    def disc(discNum=6):
        return np.array([0.003 * (11 ** (n / (discNum - 1))) for n in range(discNum)])[np.newaxis].T

    mor = MOR(diffFun, ['D'], [[0.003, 0.033]])
    pde = ADPDE(Domain1D(), diff=diffFun, vel=1.0, timeDependent=True, tInterval=[0, 2.0],
                IC=lambda x: -np.sin(pi * x), MORvar=mor)
    return VarNet(pde, layerWidth=[10, 20, 30], discNum=150, bDiscNum=75, tDiscNum=800, MORdiscScheme=disc)


def _mor_errors(evaluate, inp, d3, d4, kappa):
    out = []
    for k, tab in ((kappa[0], d3), (kappa[1], d4)):
        out.append(float(uf.l2Err(tab, evaluate(inp, k))))
    return out


@pytest.mark.gpu
def test_operator_1dtmor_tables_hip_vs_oracle():
    """Two epochs (6 kappa batches x 20 mini-batches = 240 Adam steps, in the reference's order
    VarNet.py:1350 / VarNetUtility.py:1043) on both sides from the same theta_0."""
    inp, d3, d4, kappa = tables()
    vn = _mor_setup()
    fd, eng = vn.fixData, vn.engine
    _silence(vn)
    td = vn._build_tdata(batchNum=20)
    eng.set_weights([1.0, 1.0, 1.0])
    trainW, _, _ = vn.trainWeight([10., 10., 1.], td)                     # Operator_1DtMOR.py:204
    w_e = td.towerWeights(trainW)
    eng.set_weights(w_e)
    theta0 = eng.get_params()
    epochs, q = 2, fd.integNum
    acc = torch.zeros(epochs, dtype=torch.float32, device=eng.device)
    for ep in range(epochs):
        for mb in range(fd.MORbatchNum):
            td.select_mor(mb)
            vn.optimIter(td, mb, acc[ep])
    gl = acc.cpu().numpy().astype(np.float64)
    # the oracle on the same feeds
    torch.set_num_threads(16)
    f32 = lambda t: t.cpu().numpy()
    theta = theta0.copy()
    adam = og.TF1Adam(theta.size, lr=vn.learning_rate, dtype=np.float32)
    cl = np.zeros(epochs)
    for ep in range(epochs):
        for mb in range(fd.MORbatchNum):
            d = td.mor[mb]
            for bi in range(td.batchNum):
                n0, n1 = td.block(bi)
                n = (n1 - n0) * q
                res, g = og.loss_and_grad(
                    theta, 3, [10, 20, 30], torch.float32, Input=f32(d['Input'][n0 * q:n1 * q]),
                    gcoef=f32(d['gcoef'][n0 * q:n1 * q]), source=None,
                    N=np.tile(fd.N, n1 - n0).reshape(n, 1).astype(np.float32),
                    dNt=np.tile(fd.dNt, n1 - n0).reshape(n, 1).astype(np.float32), integW=None,
                    intShape=[n1 - n0, q], detJ=float(fd.detJ), detJvec=False, biInput=f32(d['biInput']),
                    biLabel=f32(d['biLabel']).reshape(-1, 1), bDof=fd.bDofsum, biDimVal=float(fd.biDimVal), w=w_e,
                    dim=1, time_dependent=True, is_source=False, integWflag=False)
                theta = adam.step(theta, g)
                cl[ep] += res['loss']
    dev = np.abs(gl - cl) / np.abs(cl)
    ev_g = lambda X, k: vn.evaluate(x=X[:, 0:1], t=X[:, 1:2], MORarg=[[k]])          # Operator_1DtMOR.py:217
    ev_c = lambda X, k: og.forward(theta.astype(np.float64), 3, [10, 20, 30], torch.float64,
                                   np.hstack([X, k * np.ones([X.shape[0], 1])]))
    e_g, e_c = _mor_errors(ev_g, inp, d3, d4, kappa), _mor_errors(ev_c, inp, d3, d4, kappa)
    fdiff = max(float(np.max(np.abs(ev_g(inp, k) - ev_c(inp, k)))) for k in kappa)
    record('operator_1dtmor_budget', dict(adam_steps=epochs * 120, max_rel_epoch_loss_dev=float(dev.max()),
                                           l2Err_table_hip=e_g, l2Err_table_oracle=e_c,
                                           max_field_diff_at_table_points=fdiff, epoch_losses_hip=gl.tolist(),
                                           epoch_losses_oracle=cl.tolist()))
    print('MOR tables, %d steps: l2Err hip %s oracle %s, field diff %.2e, loss dev %.2e' % (epochs * 120, e_g, e_c, fdiff, dev.max()))
    assert dev.max() <= 1e-2
    assert fdiff <= FIELD_BAR
    assert max(abs(a - b) for a, b in zip(e_g, e_c)) <= 1e-4
    eng.close()


@pytest.mark.gpu
def test_operator_1dtmor_tables_long_run(tmp_path):
    """Operator_1DtMOR.py:204 as the script runs it (uniform sampling, 20 shuffled mini-batches per kappa,
    saveMORdata), bounded to 1500 epochs = 180 000 Adam steps; then Operator_1DtMOR.py:214-224 with the tables
    paired with the diffusivity they were tabulated for."""
    inp, d3, d4, kappa = tables()
    vn = _mor_setup()
    np.random.seed(0)
    res = vn.train(str(tmp_path), weight=[10., 10., 1.], smpScheme='uniform', saveMORdata=True, batchNum=20,
                   shuffleData=True, epochNum=1500, saveFreq=500, verbose=False)
    ev = lambda X, k: vn.evaluate(x=X[:, 0:1], t=X[:, 1:2], MORarg=[[k]])
    e = _mor_errors(ev, inp, d3, d4, kappa)
    swapped = [float(uf.l2Err(d4, ev(inp, kappa[0]))), float(uf.l2Err(d3, ev(inp, kappa[1])))]
    record('operator_1dtmor_long', dict(epochs=len(res.lossAll), l2Err_table_hip=e, l2Err_with_the_scripts_swapped_pairing=swapped,
                                         loss_first_last=[float(res.lossAll[0]), float(res.lossAll[-1])]))
    print('MOR tables, long run: l2Err kappa=0.01/pi %.5f, kappa=0.005 %.5f (swapped pairing: %.5f %.5f)' % (e[0], e[1], swapped[0], swapped[1]))
    assert max(e) <= TABLE_BAR_MOR
    vn.engine.close()


# ---------------------------------------------------------------------------------------------------------
# Operator_2Dt.py:89-132: the reference's analytical solution of its 2D+t demo (Leij & Dane, integrated over time), the
# known answer behind BASELINE config 3's problem.  tests/golden/cexfun_2dt.npz = outputs of the reference's own function
# (oracle/gen_golden_cexfun.py); the metric is the script's: l2Err over ALL 151 time nodes (Operator_2Dt.py:174-183).
GOLD2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cexfun_2dt.npz')
TABLE_BAR_2DT = 0.40        # l2Err(cExFun, HIP field) over all 151 time nodes, down-scaled Operator_2Dt ([20,10] grid), 6000 epochs: measured 0.307


def test_cexfun_restatement_matches_reference_outputs():
    import importlib.util
    spec = importlib.util.spec_from_file_location('ex_operator_2dt', os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'examples', 'operator_2dt.py'))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    g = np.load(GOLD2)
    np.testing.assert_allclose(g['params'], [ex.T, ex.q[0], ex.q[1], ex.kappa, ex.c0, ex.a, ex.nt])
    np.testing.assert_allclose(ex.cExFun(g['x']), g['c_all'], rtol=1e-12, atol=1e-14)
    inlet = (g['x'][:, 0] < 1e-4) & (np.abs(g['x'][:, 1]) < 0.2)
    assert inlet.sum() >= 3 and np.all(g['c_all'][inlet] == 1.0) and np.all(g['c_all'][:, 0][~inlet] == 0.0)


def _l2err_2dt(forward, g):
    """Operator_2Dt.py:174-183: cEx = cExFun(coord) flattened row-major, cApp on pairMats(coord, tcoord)."""
    x, T, nt = g['x'], float(g['params'][0]), int(g['params'][6])
    tcoord = np.linspace(0, T, num=nt).reshape(nt, 1)
    Input = uf.pairMats(x, tcoord)
    return float(uf.l2Err(g['c_all'].reshape(-1, 1), forward(Input)))


@pytest.mark.gpu
def test_operator_2dt_analytic_hip_vs_oracle_and_long_run(tmp_path):
    from tests.test_operator_parity_gpu import run_both
    from tests.test_varnet_gpu import op2dt
    g = np.load(GOLD2)
    vn = op2dt([10, 20], [20, 10], 10, 15)                             # Operator_2Dt.py:136-158, grid scaled down
    gl, cl, th_g, th_c, _ = run_both(vn, [5., 1., 1.], 400)          # weight: Operator_2Dt.py:167
    e_g = _l2err_2dt(lambda X: vn.evaluate(x=X[:, :2], t=X[:, 2:3]), g)
    e_c = _l2err_2dt(lambda X: og.forward(th_c.astype(np.float64), 3, [10, 20], torch.float64, X), g)
    dev = float(np.max(np.abs(gl - cl) / np.abs(cl)))
    assert dev <= 1e-2 and abs(e_g - e_c) <= 1e-4, (dev, e_g, e_c)
    vn.engine.init_params(seed=0)
    res = vn.train(str(tmp_path), weight=[5., 1., 1.], smpScheme='uniform', epochNum=6000, saveFreq=2000, verbose=False, lossLag=16)
    e_long = _l2err_2dt(lambda X: vn.evaluate(x=X[:, :2], t=X[:, 2:3]), g)
    record('operator_2dt_downscaled', dict(budget_steps=400, max_rel_loss_dev=dev, l2Err_cExFun_hip=e_g, l2Err_cExFun_oracle=e_c,
                                            long_epochs=len(res.lossAll), l2Err_cExFun_hip_long=e_long,
                                            loss_first_last=[float(res.lossAll[0]), float(res.lossAll[-1])]))
    print('2Dt analytic: 400 steps l2Err hip %.5f oracle %.5f (loss dev %.1e); %d epochs: %.5f' % (e_g, e_c, dev, len(res.lossAll), e_long))
    assert e_long <= TABLE_BAR_2DT
    vn.engine.close()
