"""
Trajectory and field parity on the reference's own Operator inputs (north_star: "trained fields and loss
trajectories match the reference ... on the same Operator_1Dt/2Dt inputs to a stated fp32 tolerance"), and
full-size checks of BASELINE configs 2 and 5.

The checker is the oracle: the fp32 PyTorch-CPU autograd restatement of the TF-1 graph + TF-1 Adam
(oracle/tf1_graph.py), run from the same theta_0 on the same problem-layer inputs.  Tolerances:
max relative loss deviation over the trajectory <= 1e-2; l2Err against the exact solution within 10 % of the
oracle run's; trained fields compared directly as well.
"""
import numpy as np
import pytest
import torch

from oracle import tf1_graph as og
from tests.test_varnet_host import cExact, pi
from tests.test_varnet_gpu import op1dt, op2dt
from varnet_amd import ADPDE, Domain1D, MOR, VarNet
from varnet_amd.utility import UF

pytestmark = pytest.mark.gpu
uf = UF()


def oracle_kwargs(vn, td, w, rows=None):
    fd, d = vn.fixData, td.mor[0]
    q = fd.integNum
    n_k = fd.nt if rows is None else rows
    n = n_k * q
    src = d['source']
    return dict(Input=d['Input'][:n].cpu().numpy(), gcoef=d['gcoef'][:n].cpu().numpy(),
                source=None if src is None else src[:n].cpu().numpy().reshape(n, 1),
                N=np.tile(fd.N, n_k).reshape(n, 1).astype(np.float32),
                dNt=np.tile(fd.dNt, n_k).reshape(n, 1).astype(np.float32),
                integW=None if fd.integW is None else np.reshape(fd.integW, (1, q)).astype(np.float32),
                intShape=[n_k, q], detJ=float(fd.detJ), detJvec=False,
                biInput=d['biInput'].cpu().numpy(), biLabel=d['biLabel'].cpu().numpy().reshape(-1, 1),
                bDof=fd.bDofsum, biDimVal=float(fd.biDimVal), w=np.asarray(w, dtype=float), dim=vn.dim,
                time_dependent=vn.PDE.timeDependent, is_source=vn.lossOpt['isSource'],
                integWflag=vn.lossOpt['integWflag'])


def run_both(vn, weight, steps, threads=16):
    """`steps` Adam steps on the HIP engine and on the fp32 oracle from the same theta_0, same inputs, same
    weights (trainWeight's default branch evaluated on the device).  Returns losses and final parameters."""
    eng = vn.engine
    td = vn._build_tdata()
    td.select_mor(0)

    class Log:
        verbose = False

        def writeCase(self, s):
            pass
    vn.trainRes = Log()
    eng.set_weights([1.0, 1.0, 1.0])
    trainW, _, _ = vn.trainWeight(weight, td)
    eng.set_weights(trainW)
    theta0 = eng.get_params()
    lg = torch.zeros(1, dtype=torch.float32, device=eng.device)
    gl = []
    for _ in range(steps):
        eng.train_step(0, lg)
        gl.append(float(lg.item()))
    theta_g = eng.get_params()
    torch.set_num_threads(threads)
    kw = oracle_kwargs(vn, td, trainW)
    theta = theta0.copy()
    adam = og.TF1Adam(theta.size, lr=vn.learning_rate, dtype=np.float32)
    cl = []
    for _ in range(steps):
        res, g = og.loss_and_grad(theta, vn.inpDim, vn.layerWidth, torch.float32, **kw)
        theta = adam.step(theta, g)
        cl.append(res['loss'])
    return np.array(gl), np.array(cl), theta_g, theta, trainW


def test_operator_1dt_trajectory_and_field_parity():
    """BASELINE config 1: Operator_1Dt.py:144-186 problem, 3x20 MLP, discNum=20, tDiscNum=300 (96 000 points),
    weight [10,10,1]; 1000 Adam steps."""
    vn = op1dt([20, 20, 20], 20, 300, cEx=cExact)
    assert vn.fixData.nT == 96000
    gl, cl, th_g, th_c, trainW = run_both(vn, [10., 10., 1.], 1000)
    np.testing.assert_allclose(gl[0], 1e6, rtol=1e-4)                 # trainWeight: initial weighted loss = 1e6
    dev = np.abs(gl - cl) / np.abs(cl)
    assert dev.max() <= 1e-2, dev.max()
    assert cl[-1] < 0.5 * cl[0] and gl[-1] < 0.5 * gl[0]
    # fields on uniform_input: HIP-trained network vs oracle-trained network, and both against the exact solution
    ui, cEx = vn.fixData.uniform_input, vn.fixData.cEx
    u_g = vn.evaluate()
    u_c = og.forward(th_c.astype(np.float64), 2, [20, 20, 20], torch.float64, ui)
    e_g, e_c = uf.l2Err(cEx, u_g), uf.l2Err(cEx, u_c)
    assert abs(e_g - e_c) <= 0.10 * e_c, (e_g, e_c)
    assert uf.l2Err(u_c, u_g) <= 1e-2
    _, _, err, _ = vn.residual()
    np.testing.assert_allclose(err, e_g, rtol=1e-6)                    # VarNet.residual reports the same l2Err
    print('cfg1: max rel loss dev %.2e, l2Err hip %.4f oracle %.4f, field diff %.2e'
          % (dev.max(), e_g, e_c, uf.l2Err(u_c, u_g)))
    vn.engine.close()


def test_operator_2dt_downscaled_trajectory_and_field_parity():
    """Operator_2Dt.py:136-167 problem and network ([10,20], weight [5,1,1]), grid scaled down to
    discNum=[20,10], bDiscNum=10, tDiscNum=15 (192 000 points); 400 Adam steps."""
    vn = op2dt([10, 20], [20, 10], 10, 15)
    assert vn.fixData.integNum == 64
    gl, cl, th_g, th_c, trainW = run_both(vn, [5., 1., 1.], 400)
    dev = np.abs(gl - cl) / np.abs(cl)
    assert dev.max() <= 1e-2, dev.max()
    ui = vn.fixData.uniform_input
    u_g = vn.evaluate()
    u_c = og.forward(th_c.astype(np.float64), 3, [10, 20], torch.float64, ui)
    assert uf.l2Err(u_c, u_g) <= 1e-2
    print('2dt: max rel loss dev %.2e, field diff %.2e' % (dev.max(), uf.l2Err(u_c, u_g)))
    vn.engine.close()


# ---- BASELINE config 2 at full size -----------------------------------------------------------------------
@pytest.fixture(scope='module')
def cfg2():
    vn = op1dt([50] * 4, 50, 200)                    # 10 000 test functions x 16 = 160 000 points, no source
    td = vn._build_tdata()
    td.select_mor(0)
    vn.engine.set_weights([3.0, 2.0, 5.0])
    yield vn, td
    vn.engine.close()


def _grad(eng, batch=0):
    gb = eng.bind_grad_buffer()
    eng.grad(batch)
    torch.cuda.synchronize()
    return gb.cpu().numpy().astype(np.float64)


def test_config2_fullsize(cfg2):
    from varnet_amd.engine import VNEngine
    vn, td = cfg2
    fd, eng = vn.fixData, vn.engine
    assert (fd.nt, fd.nT, fd.integNum, eng.P) == (10000, 160000, 16, 7851) and not vn.lossOpt['isSource']
    assert eng.kernel_path()[0] == 3                                  # the 8-wave fused kernel
    g1, g2 = _grad(eng), _grad(eng)
    assert np.array_equal(g1, g2)                                     # deterministic
    d = td.mor[0]
    gen = VNEngine(1, 2, [50] * 4, True, 16, kernel=1)                # independent generic kernels
    gen.set_params(eng.get_params())
    gen.set_fe_table(fd.N, fd.dNt)
    gen.set_interior(0, d['Input'], d['gcoef'], None, n_k=fd.nt, detJ=fd.detJ)
    gen.set_bic(d['biInput'], d['biLabel'], fd.bDofsum, fd.biDimVal)
    gen.set_weights([3.0, 2.0, 5.0])
    gg = _grad(gen)
    P = eng.P
    assert np.max(np.abs(g1[:P] - gg[:P])) <= 2e-4 * np.max(np.abs(gg[:P]))
    assert abs(g1[P] - gg[P]) <= 1e-4 * abs(gg[P])
    gen.close()
    # sampled oracle check: the first 300 test functions (fp64 oracle)
    q = fd.integNum
    eng.set_interior(1, d['Input'][:300 * q], d['gcoef'][:300 * q], None, n_k=300, detJ=fd.detJ)
    gs = _grad(eng, 1)
    kw = oracle_kwargs(vn, td, [3.0, 2.0, 5.0], rows=300)
    kw = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v) for k, v in kw.items()}
    ref, gref = og.loss_and_grad(eng.get_params().astype(np.float64), 2, [50] * 4, torch.float64, **kw)
    assert abs(gs[P] - ref['loss']) <= 1e-5 * abs(ref['loss'])
    assert np.max(np.abs(gs[:P] - gref)) <= 1e-4 * np.max(np.abs(gref))
    # shard additivity (what the towers rely on)
    half = fd.nt // 2
    eng.set_interior(2, d['Input'][:half * q], d['gcoef'][:half * q], None, n_k=half, detJ=fd.detJ)
    eng.set_interior(3, d['Input'][half * q:], d['gcoef'][half * q:], None, n_k=fd.nt - half, detJ=fd.detJ)
    eng.set_weights([1.5, 1.0, 5.0])
    ga, gb_ = _grad(eng, 2), _grad(eng, 3)
    assert np.max(np.abs(ga[:P] + gb_[:P] - g1[:P])) <= 2e-5 * np.max(np.abs(g1[:P]))
    assert abs(ga[P] + gb_[P] - g1[P]) <= 2e-5 * abs(g1[P])
    eng.set_weights([3.0, 2.0, 5.0])
    # the de-duplicated formulation at config-2 size (bench.py extra.config2_small_step.dedup): 41 004 unique points for 160 000
    # rows (3.9 rows per point; dim = 1: 8 F_pt per unique point against 6 F_pt per row), same loss and gradient as the row-wise
    # launch and as the independent generic kernels, bitwise reproducible, and switching it off restores the row-wise bits
    U = td.enable_dedup()
    assert U == 41004
    gd, gd2 = _grad(eng), _grad(eng)
    assert np.array_equal(gd, gd2)
    assert np.max(np.abs(gd[:P] - g1[:P])) <= 1e-4 * np.max(np.abs(g1[:P]))
    assert np.max(np.abs(gd[:P] - gg[:P])) <= 2e-4 * np.max(np.abs(gg[:P]))
    assert np.allclose(gd[P:P + 4], g1[P:P + 4], rtol=1e-5)
    td.disable_dedup()
    assert np.array_equal(_grad(eng), g1)


# ---- BASELINE config 5 at the Operator_1DtMOR sizes -----------------------------------------------------
def test_config5_fullsize_fp64_residual_and_mor_batch():
    """Operator_1DtMOR.py:166-204: kappa in 6 log-spaced values as third network input, [10,20,30] net,
    discNum=150, tDiscNum=800, batchNum=20: fp64 residual on uniform_input vs the fp64 oracle on a 5 000-row
    sample per kappa (<= 1e-10), and one MOR mini-batch's loss / gradient vs the fp64 oracle."""
    def diffFun(x, t=0, D=0.01):
        return D * np.ones([np.shape(x)[0], 1])

    def disc(discNum=6):
        return np.array([0.003 * (11 ** (n / (discNum - 1))) for n in range(discNum)])[np.newaxis].T

    mor = MOR(diffFun, ['D'], [[0.003, 0.033]])
    pde = ADPDE(Domain1D(), diff=diffFun, vel=1.0, timeDependent=True, tInterval=[0, 2.0],
                IC=lambda x: -np.sin(pi * x), MORvar=mor)
    vn = VarNet(pde, layerWidth=[10, 20, 30], discNum=150, bDiscNum=75, tDiscNum=800, MORdiscScheme=disc)
    fd, eng = vn.fixData, vn.engine
    assert fd.nt == 120000 and fd.MORbatchNum == 6 and vn.inpDim == 3 and eng.P == 921
    # a few Adam steps so that the parameters are not the symmetric initial ones
    td = vn._build_tdata(batchNum=20)
    assert td.batchNum == 20 and td.batchLen == 6000
    eng.set_weights(td.towerWeights([1e3, 1e3, 1.0]))
    acc = torch.zeros((), dtype=torch.float32, device=eng.device)
    for mb in range(fd.MORbatchNum):
        td.select_mor(mb)
        vn.optimIter(td, mb, acc)
    torch.cuda.synchronize()
    assert eng.step == 6 * 20 and np.isfinite(float(acc.item()))
    flat = eng.get_params().astype(np.float64)
    ui = fd.uniform_input
    assert ui.shape[0] == 120000
    rng = np.random.default_rng(0)
    for b in range(fd.MORbatchNum):
        r, resVec, _, cApp = vn.residual(batch=b, fp64=True)
        idx = np.sort(rng.choice(ui.shape[0], 5000, replace=False))
        cols, _, inpArg = vn._mor_columns(b, ui.shape[0])
        diff, vel, src = vn.PDEinpData(ui, inpArg)
        X = np.hstack([ui, cols])[idx]
        uref, rref = og.residual(flat, 3, [10, 20, 30], torch.float64, X, diff[idx], vel[idx], src[idx],
                                 fd.d_diff[idx], 1, True)
        assert np.max(np.abs(cApp[idx] - uref)) <= 1e-10 * max(1.0, np.max(np.abs(uref)))
        assert np.max(np.abs(resVec[idx] - rref)) <= 1e-10 * max(1.0, np.max(np.abs(rref)))
        assert np.isfinite(r)
    # one MOR mini-batch (kappa batch 4, mini-batch 7) against the fp64 oracle
    mb, bi = 4, 7
    td.select_mor(mb)
    w_e = td.towerWeights([1e3, 1e3, 1.0])
    eng.set_weights(w_e)
    gb = eng.bind_grad_buffer()
    eng.grad(td.engine_batch(mb, bi))
    torch.cuda.synchronize()
    g = gb.cpu().numpy().astype(np.float64)
    d = td.mor[mb]
    q = fd.integNum
    n0, n1 = td.block(bi)
    n = (n1 - n0) * q
    f64 = lambda t: t.cpu().numpy().astype(np.float64)
    ref, gref = og.loss_and_grad(
        eng.get_params().astype(np.float64), 3, [10, 20, 30], torch.float64,
        Input=f64(d['Input'][n0 * q:n1 * q]), gcoef=f64(d['gcoef'][n0 * q:n1 * q]), source=None,
        N=np.tile(fd.N, n1 - n0).reshape(n, 1).astype(np.float32).astype(np.float64),
        dNt=np.tile(fd.dNt, n1 - n0).reshape(n, 1).astype(np.float32).astype(np.float64), integW=None,
        intShape=[n1 - n0, q], detJ=float(np.float32(fd.detJ)), detJvec=False, biInput=f64(d['biInput']),
        biLabel=f64(d['biLabel']).reshape(-1, 1), bDof=fd.bDofsum, biDimVal=float(fd.biDimVal), w=w_e, dim=1,
        time_dependent=True, is_source=False, integWflag=False)
    P = eng.P
    assert abs(g[P] - ref['loss']) <= 1e-5 * abs(ref['loss'])
    assert np.max(np.abs(g[:P] - gref)) <= 1e-4 * np.max(np.abs(gref))
    eng.close()
