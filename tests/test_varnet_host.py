"""
CPU tests of the host logic of `VarNet` with the oracle-backed test engine
(tests/oracle_engine.py): data assembly known answers recorded in SURVEY.md 8(c) from the
reference's own code, the analytic-solution known answer, batching / sharding rules, the
training loop, checkpoints.
"""
import os
import numpy as np
import pytest

from varnet_amd.domain import Domain1D, PolygonDomain2D
from varnet_amd.adpde import ADPDE
from varnet_amd.mor import MOR
from varnet_amd import varnet as vmod
from varnet_amd.varnet import VarNet
from tests.oracle_engine import OracleEngine

pi = np.pi


@pytest.fixture(autouse=True)
def cpu_engine(monkeypatch):
    def make(self, processors):
        fd = self.fixData
        return OracleEngine(self.dim, self.inpDim, self.layerWidth, self.PDE.timeDependent, fd.integNum,
                            isSource=self.lossOpt['isSource'], integWflag=self.lossOpt['integWflag'],
                            learning_rate=self.learning_rate)
    monkeypatch.setattr(VarNet, '_make_engine', make)


def cExact(x, t, trunc=800, u=1.0, D=0.1 / pi, deriv=False):
    """Fourier-series solution of the 1D+t problem (Operator_1Dt.py:78-108); with deriv=True also
    its analytic x-derivative."""
    p = np.arange(0, trunc + 1.0).reshape(1, trunc + 1)
    c0 = 16 * pi ** 2 * D ** 3 * u * np.exp(u / D / 2 * (x - u * t / 2))
    e1 = np.exp(-D * p ** 2 * pi ** 2 * t)
    e2 = np.exp(-D * (2 * p + 1) ** 2 * pi ** 2 * t / 4)
    c1d = u ** 4 + 8 * (u * pi * D) ** 2 * (p ** 2 + 1) + 16 * (pi * D) ** 4 * (p ** 2 - 1) ** 2
    c2d = u ** 4 + (u * pi * D) ** 2 * (8 * p ** 2 + 8 * p + 10) + (pi * D) ** 4 * (4 * p ** 2 + 4 * p - 3) ** 2
    sh, ch = np.sinh(u / D / 2), np.cosh(u / D / 2)
    S = sh * np.sum((-1) ** p * 2 * p * np.sin(p * pi * x) * e1 / c1d, axis=-1, keepdims=True) + \
        ch * np.sum((-1) ** p * (2 * p + 1) * np.cos((p + 0.5) * pi * x) * e2 / c2d, axis=-1, keepdims=True)
    c = c0 * S
    ind0 = t == 0
    c[ind0] = -np.sin(pi * x[ind0])
    if not deriv:
        return c
    Sx = sh * np.sum((-1) ** p * 2 * p * p * pi * np.cos(p * pi * x) * e1 / c1d, axis=-1, keepdims=True) - \
        ch * np.sum((-1) ** p * (2 * p + 1) * (p + 0.5) * pi * np.sin((p + 0.5) * pi * x) * e2 / c2d,
                    axis=-1, keepdims=True)
    return c, c0 * (u / D / 2 * S + Sx)


def op1dt(layerWidth=[20], discNum=20, tDiscNum=300, cEx=None):
    pde = ADPDE(Domain1D(), diff=0.1 / pi, vel=1.0, timeDependent=True, tInterval=[0, 2.0],
                IC=lambda x: -np.sin(pi * x), cEx=cEx)
    return VarNet(pde, layerWidth=layerWidth, discNum=discNum, bDiscNum=None, tDiscNum=tDiscNum)


def op2dt(discNum=[80, 40], bDiscNum=40, tDiscNum=75, layerWidth=[10, 20]):
    verts = np.array([[0.0, -0.5], [0.0, -0.2], [0.0, 0.2], [0.0, 0.5], [2.0, 0.5], [2.0, -0.5]])
    BC = [[], [0.0, 1.0, 1.0], [], [], [], []]
    pde = ADPDE(PolygonDomain2D(verts), diff=1e-3, vel=[1., 0.], tInterval=[0, 1.5], BCs=BC, IC=0.0)
    return VarNet(pde, layerWidth=layerWidth, discNum=discNum, bDiscNum=bDiscNum, tDiscNum=tDiscNum)


def test_survey_known_answers_1dt():
    """Values the survey recorded by running the reference's own FIXData / trainingPoints on the
    Operator_1Dt settings."""
    vn = op1dt()
    fd = vn.fixData
    assert (fd.nt, fd.nT, fd.integNum) == (6000, 96000, 16)
    np.testing.assert_allclose(fd.detJ, 1.5873015873e-4, rtol=1e-10)
    assert list(fd.biDof) == [300, 300, 20] and fd.bDofsum == 600
    assert fd.biDimVal == 2.0
    np.testing.assert_allclose(fd.hVec.reshape(-1), [0.0952381, 0.00666667], rtol=1e-6)
    Input, _, biInput, biDof = vn.trainingPoints()
    assert Input.shape == (96000, 2) and biInput.shape == (620, 2)
    np.testing.assert_allclose(Input[0], [-0.88463573, 0.0080755], atol=5e-9)
    np.testing.assert_allclose([Input[:, 1].min(), Input[:, 1].max()], [0.00141, 2.00526], atol=5e-6)
    assert vn.lossOpt == {'integWflag': False, 'isSource': False}
    assert vn.inpDim == 2


def test_survey_known_answers_2dt():
    vn = op2dt()
    fd = vn.fixData
    assert (fd.nt, fd.nT, fd.integNum) == (240000, 15360000, 64)
    np.testing.assert_allclose(fd.detJ, 1.50557e-6, rtol=1e-5)
    assert list(fd.biDof) == [900, 1200, 900, 6000, 3000, 6000, 3200]


def test_training_point_layout_small_2dt():
    vn = op2dt(discNum=[4, 3], bDiscNum=3, tDiscNum=4)
    fd = vn.fixData
    Input, _, biInput, biDof = vn.trainingPoints()
    q = fd.integNum
    mesh = vn.PDE.domain.getMesh([4, 3], 3)
    ht, t = vn.timeDisc()
    # row r = k*q + p; test functions space-major, time-minor
    for k in (0, 5, fd.nt - 1):
        xk = mesh.coordinates[k // 4]
        tk = t[k % 4, 0]
        for p in (0, 7, q - 1):
            exp = np.concatenate([xk + mesh.he * fd.delta[:2, p], [tk + ht * fd.delta[2, p]]])
            np.testing.assert_allclose(Input[k * q + p], exp, rtol=1e-13)
    assert biInput.shape[0] == sum(biDof)
    lab = vn.biTrainData(biInput, biDof)
    # the only non-zero Dirichlet data is c=1 on edge 1 (x=0,|y|<0.2)
    o = biDof[0]
    assert np.all(lab[o:o + biDof[1]] == 1.0) and np.all(lab[:o] == 0.0) and np.all(lab[o + biDof[1]:] == 0.0)


def test_exact_solution_weak_residual_is_small():
    """SURVEY 8c known answer: assembled with the tables, the weak residual of the analytic 1D+t
    solution is ~0 (varLoss ~5e-3) against ~71 for the frozen initial condition."""
    vn = op1dt()
    fd = vn.fixData
    Input, _, _, _ = vn.trainingPoints()
    nt, q = fd.nt, fd.integNum
    x, t = Input[:, 0:1], Input[:, 1:2]
    D, u = 0.1 / pi, 1.0
    N, dNx, dNt = fd.N, fd.dNx[:, 0], fd.dNt

    def var_loss(c, cx):
        gcoef = (D * np.tile(dNx, nt) + u * np.tile(N, nt))
        int1 = cx[:, 0] * gcoef - c[:, 0] * np.tile(dNt, nt)
        R = int1.reshape(nt, q).sum(axis=1)
        return fd.detJ * np.sum(R ** 2), np.abs(R).max()

    c, cx = cExact(x, t, deriv=True)
    v_exact, rmax = var_loss(c, cx)
    c0 = -np.sin(pi * x)
    v_ic, _ = var_loss(c0, -pi * np.cos(pi * x))
    assert v_exact < 2e-2 and rmax < 1.0
    assert 60 < v_ic < 80


def test_batching_and_weights_rule():
    vn = op1dt(discNum=5, tDiscNum=6)
    td = vn._build_tdata(batchNum=4)
    assert td.batchLen == int(np.ceil(30 / 4)) and td.batchNum == 4
    blocks = [td.block(b) for b in range(4)]
    assert blocks == [(0, 8), (8, 16), (16, 24), (24, 30)]
    td2 = vn._build_tdata(batchLen=7)
    assert td2.batchNum == int(np.ceil(30 / 7)) and td2.block(td2.batchNum - 1)[1] == 30
    with pytest.raises(ValueError):
        vn._build_tdata(batchNum=2, batchLen=3)


def test_train_loop_decreases_loss_and_checkpoints(tmp_path):
    vn = op1dt(layerWidth=[8, 8], discNum=6, tDiscNum=8, cEx=cExact)
    res = vn.train(str(tmp_path), weight=[10., 10., 1.], epochNum=40, saveFreq=20, verbose=False)
    assert len(res.lossAll) == 40 and res.lossAll[-1] < res.lossAll[0]
    np.testing.assert_allclose(res.lossAll[0], 1e6, rtol=1e-6)       # initial weighted loss normalised to 1e6
    assert os.path.exists(os.path.join(str(tmp_path), 'caseData.txt'))
    assert os.path.exists(os.path.join(str(tmp_path), 'trainData.vn'))
    p = vn.engine.get_params().copy()
    vn.engine.init_params(seed=5)
    n = vn.loadModel()
    assert n in (20, 40)
    c = vn.evaluate()
    assert c.shape == (vn.fixData.uniform_input.shape[0], 1)
    r, rv, err, ca = vn.residual()
    assert np.isfinite(r) and np.isfinite(err) and rv.shape == c.shape
    layers = vn.saveNNparam(dpOut=True, matOut=True, timeFirst=True)
    assert layers[0][0].shape == (8, 2) and layers[0][1].shape == (8, 1) and layers[2][0].shape == (1, 8)
    flat = vn.engine.get_params()
    W1 = flat[:16].reshape(2, 8).T
    np.testing.assert_array_equal(layers[0][0][:, 0], W1[:, 1])          # timeFirst: t column first
    np.testing.assert_array_equal(layers[0][0][:, 1], W1[:, 0])
    import scipy.io as spio
    npdir = os.path.join(str(tmp_path), 'NN_parameters')
    np.testing.assert_array_equal(spio.loadmat(os.path.join(npdir, 'W1.mat'))['W1'], layers[0][0])
    np.testing.assert_array_equal(spio.loadmat(os.path.join(npdir, 'B3.mat'))['B3'], layers[2][1])
    txt = open(os.path.join(npdir, 'W2.m')).read()
    assert txt.startswith("datatype = 'real';") and 'nrows = 8; ncolumns = 8;' in txt and 'W2(8,8) = ' in txt
    assert 'length = 8;' in open(os.path.join(npdir, 'B1.m')).read()


def test_result_files_follow_reference_formats(tmp_path):
    """caseData.txt sections, trainData.vn keys and the sampling policy of TrainResult
    (VarNetUtility.py:1217-1464, 1512-1556, 1560-1622)."""
    import pickle
    vn = op1dt(layerWidth=[8, 8], discNum=6, tDiscNum=8, cEx=cExact)
    res = vn.train(str(tmp_path), weight=[1., 1., 1.], epochNum=25, tol=1e-12, verbose=False, saveFreq=10)
    case = open(os.path.join(str(tmp_path), 'caseData.txt')).read()
    order = ['VarNet Library', 'Simulation date: ', '1D time-dependent Advection-Diffusion problem without model-order-reduction.',
             'Boundary condition information:', 'Neural Network architecture:', '\tnumber of inputs: 2',
             '\ttotal number of trainable parameters: %d' % vn.engine.P, 'Processor information:', 'Optimizer information:',
             '\ttype: Adam stochastic gradient descent algorithm', 'Space-time discretization information:',
             '\tnumber of training points: %d' % vn.fixData.nt, 'Sampling scheme for training points: uniform',
             'Weighting information:', '\trequested weights: [1.0, 1.0, 1.0]', 'Stopping criteria:',
             '\tmaximum number of epochs: 25', 'Training iterations:', 'Epoch  1: loss = ', 'Epoch  9: loss = ',
             'Epoch 10: loss = ', 'Epoch 20: loss = ']
    pos = [case.index(t) for t in order]
    assert pos == sorted(pos)
    assert 'Epoch 11:' not in case and 'Batch-optimization information' not in case
    # sampled histories
    assert res.iterSmp == [10, 20] and len(res.loss) == 2 and len(res.lossAll) == 25
    assert len(res.lossComp) == 3 and len(res.residual) == 2 and len(res.error) == 2
    assert res.avgtime0 is not None and res.avgtime > 0
    # pickle schema
    with open(os.path.join(str(tmp_path), 'trainData.vn'), 'rb') as f:
        dump = pickle.load(f)
    for key in ['casepath', 'plotpath', 'caseSimline', 'saveFreq', 'verbose', 'pltReplace', 'trainWeight', 'loss', 'lossComp',
                'avgtime0', 'avgtime', 'residual', 'iterSmp', 'inpIter', 'error', 'lossVec', 'option_stopping',
                'option_trainPoint', 'option_weighting', 'option_batchOptim']:
        assert key in dump, key
    assert dump['option_stopping'] == {'epochNum': 25, 'tol': 1e-12} and dump['option_trainPoint']['smpScheme'] == 'uniform'
    assert dump['iterSmp'] == [10, 20] and os.path.isdir(dump['plotpath'])
    # comments go above the iteration log
    res.writeComment('post-run note')
    lines = open(os.path.join(str(tmp_path), 'caseData.txt')).read().split('\n')
    assert lines.index('post-run note') < lines.index('Training iterations:')
    with pytest.raises(ValueError):
        res.writeCase(3)
    # a fresh object restores the record
    from varnet_amd.varnet import TrainResult
    tr = TrainResult(str(tmp_path))
    tr.loadData()
    assert tr.iterSmp == [10, 20] and tr.option_weighting['weight'] == [1., 1., 1.]


def test_minibatch_epoch_equals_manual_steps(tmp_path):
    """batchNum=3: BC/IC weights divided by batchNum (VarNetUtility.py:900-901), one Adam step per
    mini-batch, epoch loss = sum of pre-update batch losses (VarNetUtility.py:1043-1045)."""
    vn = op1dt(layerWidth=[6], discNum=5, tDiscNum=6)
    res = vn.train(str(tmp_path), weight=[1., 1., 1.], epochNum=2, saveFreq=100, verbose=False, batchNum=3)
    assert vn.engine.step == 6
    # the record holds the per-feed weights: the reference's trainRes.trainWeight is the array updateDictFields divided in
    # place (tests/test_assembly_golden.py::test_epoch_loop_matches_reference_uniform pins this against the reference)
    tw = res.trainWeight
    np.testing.assert_allclose(vn.engine.w, tw)
    lossVal = np.reshape(res.lossComp[0], -1)
    np.testing.assert_allclose(3 * tw[0] * lossVal[0] + 3 * tw[1] * lossVal[1] + tw[2] * lossVal[2], 1e6, rtol=1e-9)


def test_mor_pipeline(tmp_path):
    def diffFun(x, t=0, D=0.01):
        return D * np.ones([np.shape(x)[0], 1])

    def disc(discNum=3):
        return np.array([0.003 * (11 ** (n / (discNum - 1))) for n in range(discNum)])[np.newaxis].T

    mor = MOR(diffFun, ['D'], [[0.003, 0.033]])
    pde = ADPDE(Domain1D(), diff=diffFun, vel=1.0, timeDependent=True, tInterval=[0, 2.0],
                IC=lambda x: -np.sin(pi * x), MORvar=mor)
    vn = VarNet(pde, layerWidth=[5, 5], discNum=5, bDiscNum=None, tDiscNum=6, MORdiscScheme=disc)
    assert vn.inpDim == 3 and vn.fixData.MORbatchNum == 3
    td = vn._build_tdata()
    for b, kappa in enumerate(disc()[:, 0]):
        Inp = td.mor[b]['Input'].numpy()
        np.testing.assert_allclose(Inp[:, 2], kappa)
        # gcoef = kappa*dNx + v*N
        g = td.mor[b]['gcoef'].numpy().reshape(vn.fixData.nt, vn.fixData.integNum)
        np.testing.assert_allclose(g[3], kappa * vn.fixData.dNx[:, 0] + vn.fixData.N, rtol=1e-12)
    res = vn.train(str(tmp_path), weight=[10., 10., 1.], epochNum=3, saveFreq=100, verbose=False)
    assert vn.engine.step == 9 and np.isfinite(res.lossAll[-1])
    c = vn.evaluate(x=np.array([[0.1], [0.2]]), t=np.array([[0.5], [0.6]]), batch=1)
    c2 = vn.evaluate(x=np.array([[0.1], [0.2]]), t=np.array([[0.5], [0.6]]), MORarg=np.array([[disc()[1, 0]]]))
    np.testing.assert_allclose(c, c2)


def test_constructor_errors():
    pde = ADPDE(Domain1D(), diff=0.1, vel=1.0, tInterval=[0, 1.0], IC=0.0)
    with pytest.raises(ValueError):
        VarNet(pde, layerWidth=[5], discNum=[3, 3], tDiscNum=4)
    with pytest.raises(ValueError):
        VarNet(pde, layerWidth=[5], discNum=3, bDiscNum=None)            # tDiscNum missing
    with pytest.raises(ValueError):
        VarNet(pde, layerWidth=5, discNum=3, bDiscNum=None, tDiscNum=4)
    with pytest.raises(NotImplementedError):
        VarNet(pde, layerWidth=[5], modelId='RNN', discNum=3, bDiscNum=None, tDiscNum=4)
    vn = VarNet(pde, layerWidth=[5], discNum=3, bDiscNum=None, tDiscNum=4)
    with pytest.raises(ValueError):
        vn.train(None, epochNum=1)
    with pytest.raises(ValueError):
        vn.train('/tmp/x', weight=[1., 1.], epochNum=1)


def test_rejection_sampling_follows_density():
    """Accepted samples follow func/max(func): density ~ x^2 on [0,1] has mean 3/4."""
    from varnet_amd.utility import UF
    uf = UF()
    np.random.seed(0)
    f = lambda x=None: (np.linspace(0, 1, 200).reshape(-1, 1) if x is None else x) ** 2
    smp = uf.rejectionSampling(f, lambda: np.random.uniform(0, 1, (200, 1)), 3000)
    assert smp.shape == (3000, 1) and abs(smp.mean() - 0.75) < 0.02
    segs = uf.listSegment(np.arange(10).reshape(10, 1), [3, 4])
    assert [len(x) for x in segs] == [3, 4, 3]
    two = uf.rejectionSampling(lambda x=None: np.ones((20, 1)) if x is None else np.ones((len(x), 1)),
                               lambda: np.random.uniform(0, 1, (20, 1)), [5, 7], [8, 12])
    assert two.shape == (12, 1)


@pytest.mark.parametrize('addTrainPts,suppFactor', [(True, 1.0), (True, 0.5), (False, 1.0)])
def test_optimal_sampling_training(tmp_path, addTrainPts, suppFactor):
    """smpScheme='optimal' (VarNet.py:1385-1421, 1696-1966): after the convergence test fires the
    training set is re-drawn from the residual field, fixed data follow updateOptimData, variables
    are re-initialised and the weights re-derived."""
    np.random.seed(1)
    vn = op1dt(layerWidth=[6, 6], discNum=6, tDiscNum=8, cEx=cExact)
    fd = vn.fixData
    nt0, q = fd.nt0, fd.integNum
    res = vn.train(str(tmp_path), weight=[10., 10., 1.], smpScheme='optimal', epochNum=12, saveFreq=5,
                   verbose=False, trainUpdelay=5, tolUpd=10.0, frac=0.5, addTrainPts=addTrainPts,
                   suppFactor=suppFactor, adjustWeight=True)
    assert res.inpIter == [5]                                      # one update (multiTrainUpd=False); the
    # convergence test reads the losses sampled every saveFreq epochs, as the reference does
    td = vn.tData
    if addTrainPts:
        nt1 = int(np.ceil(0.5 * nt0))
        assert fd.nt == nt0 + nt1 and fd.nT == fd.nt * q
        assert list(fd.biDof) == [b + int(np.ceil(0.5 * b)) for b in fd.biDof0]
        assert td.mor[0]['Input'].shape[0] == fd.nT
        assert td.mor[0]['biInput'].shape[0] == sum(fd.biDof)
    else:
        assert fd.nt == nt0 and td.mor[0]['Input'].shape[0] == nt0 * q
    if suppFactor != 1.0:
        assert fd.detJvec and np.shape(fd.detJ) == (fd.nt, 1)
        nt1 = fd.nt - nt0
        np.testing.assert_allclose(fd.detJ[:nt1, 0], fd.detJ[-1, 0] * 0.5 ** fd.feDim)
        Nr, dNxr, dNtr = fd.rows()
        assert Nr.shape == (fd.nT, 1)
        np.testing.assert_allclose(dNtr[:nt1 * q], np.tile(fd.dNt, nt1).reshape(-1, 1) / 0.5)
        assert td.mor[0]['N_rows'] is not None and td.mor[0]['detJ'].shape[0] == fd.nt
        # supports of the added test functions are half as wide
        Inp = td.mor[0]['Input'].numpy().reshape(fd.nt, q, 2)
        assert np.ptp(Inp[0, :, 0]) < 0.6 * np.ptp(Inp[-1, :, 0])
    else:
        assert not fd.detJvec
    assert vn.engine.step == 7                                     # re-initialised at epoch 5, then 7 more steps
    assert np.isfinite(res.lossAll).all()
    # weights were re-derived with 5x on BC/IC (adjustWeight) and renormalised to 1e6
    np.testing.assert_allclose(res.lossAll[5], 1e6, rtol=1e-6)


def test_simres_returns_the_fields_the_reference_plots(tmp_path):
    """simRes (VarNet.py:1970-2175) without a display: arrays on the ContourPlot grid, consistent with
    evaluate / residual / cEx, plus the reference's figure file names when plot=True."""
    vn = op1dt(layerWidth=[6, 6], discNum=6, tDiscNum=8, cEx=cExact)
    vn.train(str(tmp_path), weight=[10., 10., 1.], epochNum=4, saveFreq=2, verbose=False)
    out = vn.simRes()
    assert out['t'] == [0.0, 0.5, 1.0, 1.5, 2.0] and len(out['cApp']) == 5
    x = out['grid'].x_coord
    assert x.shape == (51, 1) and out['cApp'][2].shape == (51, 1)
    np.testing.assert_allclose(out['cApp'][2], vn.evaluate(x, 1.0), rtol=0, atol=1e-12)
    np.testing.assert_allclose(out['cEx'][3], cExact(x, 1.5 * np.ones([51, 1])), rtol=1e-12)
    np.testing.assert_allclose(out['cErr'][1], out['cEx'][1] - out['cApp'][1], rtol=0, atol=1e-12)
    _, rv, _, _ = vn.residual(np.hstack([x, 0.5 * np.ones([51, 1])]))
    np.testing.assert_allclose(out['res'][1], rv, rtol=0, atol=1e-10)
    assert len(out['l2Err']) == 5 and 'lossField' in out and out['lossField'][0].shape == (51, 1)
    vn.simRes(plot=True)
    files = set(os.listdir(os.path.join(str(tmp_path), 'plots')))
    assert {'cApp-t=0.50s.png', 'cErr.png', 'residual.png', 'lossField.png'} <= files
    with pytest.raises(ValueError):
        vn.simRes(pltFrmt='bmp')


def test_simres_2d_fields(tmp_path):
    vn = op2dt(discNum=[6, 4], bDiscNum=4, tDiscNum=3, layerWidth=[5])
    out = vn.simRes(tcoord=[0.3, 1.2])
    assert out['cApp'][0].shape == (51, 51) and out['res'][1].shape == (51, 51) and 'cEx' not in out
    cp = out['grid']
    X = np.concatenate([cp.X_coord, cp.Y_coord], axis=1)
    ref = vn.evaluate(X, 1.2).reshape(51, 51)
    ref[cp.isOutside.reshape(51, 51)] = 0.0
    np.testing.assert_allclose(out['cApp'][1], ref, rtol=0, atol=1e-12)


def test_checkpoint_uses_tf_variable_names_and_round_trips(tmp_path):
    """best_model-<n>.npz holds what tf.train.Saver stores for this graph, under the TF names
    (TFModel.py:208-242, 307): kernels [in,out], biases, Adam slots, beta powers, step."""
    vn = op1dt(layerWidth=[6, 5], discNum=5, tDiscNum=6)
    vn.train(str(tmp_path), weight=[10., 10., 1.], epochNum=6, saveFreq=2, verbose=False)
    files = sorted(f for f in os.listdir(str(tmp_path)) if f.startswith('best_model-'))
    assert len(files) <= 2 and all(f.endswith('.npz') for f in files)           # max_to_keep = 2
    z = np.load(os.path.join(str(tmp_path), files[-1]))
    names = set(z.files)
    for v in ('dense_0', 'dense_1', 'output'):
        assert {v + '/kernel', v + '/bias', v + '/kernel/Adam', v + '/kernel/Adam_1', v + '/bias/Adam_1'} <= names
    assert z['dense_0/kernel'].shape == (2, 6) and z['dense_1/kernel'].shape == (6, 5) and z['output/kernel'].shape == (5, 1)
    assert {'global_step', 'beta1_power', 'beta2_power'} <= names
    txt = open(os.path.join(str(tmp_path), 'checkpoint')).read()
    assert txt.startswith('model_checkpoint_path: ') and 'all_model_checkpoint_paths: ' in txt and 'best_model-' in txt
    before = vn.engine.export_state().copy()
    n = int(files[-1][len('best_model-'):-4])
    step_saved = int(z['global_step'])
    vn.engine.init_params(seed=9)
    assert vn.loadModel() == n and vn.engine.step == step_saved
    arr = vn.checkpoint_arrays()
    np.testing.assert_array_equal(arr['dense_1/kernel'], z['dense_1/kernel'])
    np.testing.assert_array_equal(arr['output/bias/Adam'], z['output/bias/Adam'])
    # a folder that only holds TF saver files is reported as such, not as "nothing found"
    d2 = tmp_path / 'tfonly'
    d2.mkdir()
    (d2 / 'best_model-100.index').write_bytes(b'')
    with pytest.raises(ValueError, match='TensorFlow saver files'):
        vn.loadModel(folderpath=str(d2))
    with pytest.raises(ValueError, match='no restorable'):
        d3 = tmp_path / 'empty'
        d3.mkdir()
        vn.loadModel(folderpath=str(d3))


def test_loss_lag_is_only_a_readback_schedule(tmp_path):
    """train(..., lossLag=k): k epochs per host read-back.  Same steps, same losses, same checkpoints and monitor epochs
    as lossLag=0 -- and, since round 6, the same stop: when `loss < tol` fires inside a block the engine rolls back to the
    block's start and replays up to the epoch that met the tolerance (VarNet.py:1378: the reference stops right there)."""
    runs = []
    for lag in (0, 7):
        np.random.seed(31)                           # shuffles draw from the global NumPy stream, as in the reference
        vn = op1dt(layerWidth=[6, 5], discNum=5, tDiscNum=6, cEx=cExact)
        res = vn.train(str(tmp_path / ('lag%d' % lag)), weight=[10., 10., 1.], epochNum=23, saveFreq=5, verbose=False,
                       batchNum=2, shuffleData=True, shuffleFreq=4, lossLag=lag)
        runs.append((np.array(res.lossAll), list(res.iterSmp), vn.engine.get_params().copy(), vn.engine.step,
                     sorted(f for f in os.listdir(str(tmp_path / ('lag%d' % lag))) if f.startswith('best_model'))))
    (l0, s0, p0, n0, f0), (l1, s1, p1, n1, f1) = runs
    assert n0 == n1 == 23 * 2 and s0 == s1 == [5, 10, 15, 20] and f0 == f1
    np.testing.assert_array_equal(l1, l0)
    np.testing.assert_array_equal(p1, p0)
    # the stopping test: exactly the reference's stop, whatever the read-back schedule (also the default one, lossLag=None)
    stops = []
    for lag in (0, 6, None):
        vn = op1dt(layerWidth=[6, 5], discNum=5, tDiscNum=6)
        res = vn.train(str(tmp_path / ('stop%s' % lag)), weight=[10., 10., 1.], epochNum=60, tol=0.95e6, saveFreq=100, verbose=False,
                       lossLag=lag)
        assert res.lossAll[-1] < 0.95e6 and all(v >= 0.95e6 for v in res.lossAll[:-1])
        assert vn.engine.step == len(res.lossAll)                 # not one step beyond the epoch that met the tolerance
        stops.append((np.array(res.lossAll), vn.engine.get_params().copy()))
    assert 1 < len(stops[0][0]) < 60 and (len(stops[0][0]) - 1) % 6 != 5      # the stop falls INSIDE a block of the lagged runs
    for l, p in stops[1:]:
        np.testing.assert_array_equal(l, stops[0][0])
        np.testing.assert_array_equal(p, stops[0][1])


def test_loss_lag_blocks_end_where_the_training_set_is_redrawn(tmp_path):
    """Non-uniform sampling under the read-back schedule (round 6): the re-draw test of VarNet.py:1385-1421 runs after every epoch,
    but it can first fire only at an epoch known when a block is sized (the monitors' loss samples and the epochs since the last
    re-draw decide) -- the block ends there, so the re-drawn set, the re-initialised variables and the x 5 weights start at exactly
    the epoch the reference's loop starts them.  saveFreq 4, trainUpdelay 7: the test fires at epoch 7, inside the block 5..8."""
    runs = []
    for lag in (0, None, 3):
        np.random.seed(77)
        vn = op1dt(layerWidth=[6, 5], discNum=5, tDiscNum=6, cEx=cExact)
        res = vn.train(str(tmp_path / ('lag%s' % lag)), weight=[10., 10., 1.], smpScheme='optimal', adjustWeight=True, epochNum=23,
                       saveFreq=4, trainUpdelay=7, tolUpd=1e9, verbose=False, lossLag=lag)
        runs.append((np.array(res.lossAll), list(res.inpIter), list(res.iterSmp), vn.engine.get_params().copy(), vn.engine.step,
                     np.asarray(res.trainWeight, dtype=float), vn.tData.mor[0]['Input'].shape[0]))
    l0, i0, s0, p0, n0, w0, r0 = runs[0]
    assert i0 == [7] and r0 > 5 * 6 * 16                       # re-drawn once, mid-block for both lagged schedules, points added
    for l, i, sm, p, n, w, r in runs[1:]:
        np.testing.assert_array_equal(l, l0)
        np.testing.assert_array_equal(p, p0)
        np.testing.assert_array_equal(w, w0)
        assert i == i0 and sm == s0 and n == n0 and r == r0


def test_iter_plot_writes_the_reference_files(tmp_path):
    """TrainResult.iterPlot (VarNetUtility.py:1634-1756): the five convergence plots under the reference's file names,
    refreshed by iterOutput every 10 * saveFreq epochs; the epoch prefix when pltReplace is False."""
    from varnet_amd.varnet import TrainResult
    tr = TrainResult(str(tmp_path), True, verbose=False, saveFreq=2, pltReplace=True)
    tr.trainWeight = np.array([10.0, 5.0, 1.0])
    tr.lossComp.append(np.array([1.0, 2.0, 3.0]))
    tr.inpIter = [8]
    for ep in range(1, 21):
        tr.iterOutput(ep, 100.0 / ep, 90.0 / ep, 0.01 * ep, 0.5 / ep, 0.2 / ep, np.array([1.0, 2.0, 3.0]) / ep, None)
    plots = sorted(os.listdir(tr.plotpath))
    assert plots == ['error.png', 'loss.png', 'lossComp.png', 'res_history.png', 'scaled_lossComp.png']
    assert all(os.path.getsize(os.path.join(tr.plotpath, f)) > 1000 for f in plots)
    tr.pltReplace = False
    tr.iterPlot(pltFrmt='pdf')
    assert '20_loss.pdf' in os.listdir(tr.plotpath)
    with pytest.raises(ValueError):
        tr.iterPlot(pltFrmt='bmp')


class _DedupRecorder(OracleEngine):
    """Stand-in engine that accepts `vn_set_dedup` registrations (and keeps training row-wise on the oracle): what
    `train(dedup=...)` asks of an engine can then be observed without a GPU."""
    supported = True

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.dd = {}

    def set_dedup(self, batch, Xu=None, uid=None, rowptr=None, rowidx=None):
        if Xu is None:
            self.dd.pop(batch, None)
        else:
            self.dd[batch] = (int(Xu.shape[0]), np.asarray(uid), np.asarray(rowptr), np.asarray(rowidx))

    def dedup_supported(self):
        return self.supported


@pytest.fixture
def dedup_engine(monkeypatch):
    def make(self, processors):
        fd = self.fixData
        return _DedupRecorder(self.dim, self.inpDim, self.layerWidth, self.PDE.timeDependent, fd.integNum,
                              isSource=self.lossOpt['isSource'], integWflag=self.lossOpt['integWflag'],
                              learning_rate=self.learning_rate)
    monkeypatch.setattr(VarNet, '_make_engine', make)


def test_train_chooses_the_formulation_and_says_why(tmp_path, dedup_engine, monkeypatch):
    """`train(dedup='auto')` is the default: on a uniform, unshuffled set whose step is large enough the
    de-duplicated formulation is registered for every block; where it is not, the reason is WRITTEN (caseData.txt,
    `vn.dedup_state`) and, for an explicit dedup=True, warned -- never a silent row-wise run."""
    import warnings
    kw = dict(weight=[10., 10., 1.], epochNum=2, saveFreq=100, verbose=False)
    monkeypatch.setattr(vmod, 'DEDUP_MODEL', dict(vmod.DEDUP_MODEL, fixed_us=-1e9))      # the tiny test problem "pays"
    vn = op1dt([6, 5], 5, 6)
    vn.train(str(tmp_path / 'a'), **kw)
    q = vn.fixData.integNum
    assert vn.dedup_state['on'] and vn.dedup_state['requested'] == 'auto' and list(vn.engine.dd) == [0]
    U, uid, rowptr, rowidx = vn.engine.dd[0]
    assert U == vn.dedup_state['unique_points'] == 7 * 6 * 4 and len(uid) == 30 * q     # (5+1)(6+1) elements... x 2x2 Gauss points
    assert np.array_equal(np.sort(rowidx), np.arange(30 * q)) and rowptr[-1] == 30 * q
    assert 'de-duplicated formulation, %d unique quadrature points' % U in open(str(tmp_path / 'a' / 'caseData.txt')).read()
    # two mini-batches: every block has its own map
    vn.train(str(tmp_path / 'b'), batchNum=2, **kw)
    assert sorted(vn.engine.dd) == [0, 1] and vn.dedup_state['on']
    # the real threshold: this step is far too small to pay -> 'auto' stays row-wise and says so; True forces it
    monkeypatch.undo()
    monkeypatch.setattr(VarNet, '_make_engine', lambda self, p: _DedupRecorder(
        self.dim, self.inpDim, self.layerWidth, self.PDE.timeDependent, self.fixData.integNum,
        isSource=self.lossOpt['isSource'], integWflag=self.lossOpt['integWflag'], learning_rate=self.learning_rate))
    vn = op1dt([6, 5], 5, 6)
    vn.train(str(tmp_path / 'c'), **kw)
    assert not vn.dedup_state['on'] and 'too small' in vn.dedup_state['reason'] and not vn.engine.dd
    assert 'row-wise formulation (dedup=\'auto\'): the step is too small' in open(str(tmp_path / 'c' / 'caseData.txt')).read()
    vn.train(str(tmp_path / 'd'), dedup=True, **kw)
    assert vn.dedup_state['on'] and list(vn.engine.dd) == [0]
    vn.train(str(tmp_path / 'e'), dedup=False, **kw)
    assert not vn.dedup_state['on'] and not vn.tData.dedup_on
    # every reason it cannot apply is named: shuffled mini-batches, non-uniform supports, unsupported network, dim > 3
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        vn.train(str(tmp_path / 'f'), dedup=True, batchNum=2, shuffleData=True, **kw)
    assert not vn.dedup_state['on'] and any('shuffled' in str(w.message) for w in rec)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        vn.train(str(tmp_path / 'g'), dedup='auto', batchNum=2, shuffleData=True, **kw)
    assert 'shuffled' in vn.dedup_state['reason'] and not any('dedup' in str(w.message) for w in rec)    # 'auto' records, True warns
    vn.engine.supported = False
    with pytest.warns(UserWarning, match='outside the 8-wave fused kernel family'):
        vn.train(str(tmp_path / 'h'), dedup=True, **kw)
    vn.engine.supported = True
    td = vn._build_tdata()
    vn.fixData.detJvec = True
    assert 'non-uniform' in td.dedup_applies()
    vn.fixData.detJvec = False
    td.shuffled = True
    assert td.enable_dedup() == 0 and 'shuffled' in td.dedup_reason
    with pytest.raises(ValueError, match='dedup must be'):
        vn.train(str(tmp_path / 'i'), dedup='yes', **kw)


def test_dedup_pay_off_estimate_tracks_the_real_map(dedup_engine):
    """`dedup_pays` decides before the map exists: its rows-per-point estimate stays within 15 % of the ratio the map
    then has, for whole grids and for mini-batch blocks, and BASELINE configs 2 and 3 are on the paying side."""
    from varnet_amd.varnet import unique_points
    for vn, bn in ((op1dt([20], 20, 30), None), (op1dt([20], 20, 30), 4), (op1dt([20], 50, 200), None),
                   (op2dt([8, 6], 6, 8), None), (op2dt([8, 6], 6, 8), 3), (op2dt([16, 12], 10, 12), None)):
        td = vn._build_tdata(batchNum=bn)
        fd, q = vn.fixData, vn.fixData.integNum
        n0, n1 = td.block(0)
        first = unique_points(td.mor[0]['Input_host'][n0 * q:n1 * q], fd.feDim, fd.hVec)[0]
        true = (n1 - n0) * q / len(first)
        est = td.rows_per_point_estimate()
        assert abs(est - true) <= 0.15 * true, (vn.discNum, vn.tDiscNum, bn, est, true)
    assert op1dt([50] * 4, 50, 200)._build_tdata().dedup_pays()          # config 2: 0.170 -> 0.116 ms/step measured
    assert not op1dt([6, 5], 5, 6)._build_tdata().dedup_pays()
    assert not op1dt([20] * 3, 20, 300)._build_tdata().dedup_pays()      # config 1: 36.8 us either way -> row-wise
    assert not op1dt([20], 50, 200)._build_tdata().dedup_pays()          # 160 k rows on a [20] net: 21.0 against 28.8 us
