"""
The reference-held known answers on the device path, and the chain  reference numbers -> oracle graph -> HIP engine:

  * tabulated exact values (Mojtabi & Deville) for kappa = 0.01/pi and kappa = 0.005, 25 points each
    (/root/reference/Operator_1Dt.py:113-128, /root/reference/Operator_1DtMOR.py:117-150) -> tests/golden/exact_tables.npz;
  * the Fourier-series solution `cExact` of the 1D+t problem (/root/reference/Operator_1Dt.py:78-108; the parametrised copy
    Operator_1DtMOR.py:77-110 that the MOR script evaluates at D = 0.1/pi, :226-229) -> tests/golden/cexact_1dt.npz
    (oracle/gen_golden_cexact.py: the FunctionDefs cut out of the scripts with `ast` and run);
  * the analytical solution `cExFun` of the 2D+t demo (/root/reference/Operator_2Dt.py:89-132) -> tests/golden/cexfun_2dt.npz;
  * the reference's own acceptance metric l2Err(cEx, cApp) (Operator_1Dt.py:177-186, Operator_1DtMOR.py:214-234,
    Operator_2Dt.py:174-183).

What is asserted on the GPU (numbers -> gpurun_out/r4_exact_tables.json, copied to profiles/):
  1. budget runs: the HIP engine and the fp32 oracle (the CPU restatement of TFModel.py:515-714 + TF-1 Adam), trained for the
     same fixed step budget from the same theta_0, give the same field at the reference's points (FIELD_BAR);
  2. CONVERGED runs with the scripts' own settings, the reference's metric against the reference-held answer at bars <= 0.2
     that were written down BEFORE the test's first run, from the exploration runs committed as profiles/r4_explore_*.txt
     (tools/explore_tables.py: the metric every 1000-2000 epochs along one run), and not raised afterwards;
  3. discrimination: every known answer is matched by the network evaluated at ITS diffusivity several times better than
     by the same network a decade away in kappa (see DISCRIMINATION below for what the two tables alone can and cannot tell);
  4. the oracle leg at the converged theta*: fp64 oracle loss/gradient on mini-batches vs the engine's (1e-5 / 1e-4), oracle
     forward at the reference's points vs `evaluate` (1e-6), and the ORACLE's own distance from the reference-held answer at
     the same bars -- so the bars constrain the oracle graph, not only the HIP engine.
Pairing: kappa = 0.01/pi <-> cExD3, kappa = 0.005 <-> cExD4; the reference's script swaps them
(Operator_1DtMOR.py:216-224) -- not copied.

DISCRIMINATION, stated honestly.  The two tables are 13 % apart in the metric (|d3 - d4| / |d3|), and almost all of that sits in
the boundary layer x >= 0.98, which the [10,20,30] network does not resolve to that accuracy within 60 000 epochs of the script's
settings (profiles/r4_explore_mor_60000.txt: correct pairing 0.09-0.13 / 0.04-0.16 from epoch 8000 on, the script's swapped pairing
0.06-0.13 / 0.12-0.15 -- for kappa = 0.01/pi the smoother kappa = 0.005 table is often the CLOSER one, i.e. the trained boundary
layer is too diffuse; the reference lets this run go to 500 000 epochs).  Round 5 settled it with ONE run to 120 000 epochs
(14.4 M Adam steps, 614 s; profiles/r5_explore_mor_120000.txt): for kappa = 0.005 the correct table wins from epoch ~62 000 on and by
3 x at the end (0.042 against 0.129); for kappa = 0.01/pi the smoother kappa = 0.005 table stays the closer one to the end (correct 0.103,
swapped 0.085) while the points x <= 0.9 converge to 0.013 / 0.010 -- the network's boundary layer at the larger Peclet number is still
too diffuse.  "Correct beats swapped on both" is therefore not a property the method has at any budget a test can afford, and it is
not asserted (tests/long_tables.py, not collected, asserts the half that holds).  What is asserted instead: away from the
boundary layer (x <= 0.9) the network's kappa-sensitivity u(kappa_0) - u(kappa_1) is closer to the tables' difference d3 - d4
than to zero BY A MARGIN (<= 0.9 where 1.0 means no sensitivity at all; 0.79 at this test's 12 000 epochs, <= 0.69 from epoch 18 000
on); and each answer rejects the network evaluated a decade away in kappa by a factor >= 2.5.
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

# ---- bars.  FIELD_BAR is a parity bar (HIP vs oracle).  The others are distances from reference-held answers in the reference's
# own metric (1.0 = the zero function), every one <= 0.2, fixed before the tests' first run from the exploration runs
# (profiles/r4_explore_*.txt); the measured values are in profiles/r4_exact_tables.json.
FIELD_BAR = 1e-4            # max |u_hip - u_oracle| at the reference's points (values are O(1))
FWD_BAR = 5e-6              # fp32 engine forward vs fp64 oracle forward at the SAME parameters (the bar of test_train_1dt_end_to_end; a
                            # converged net has steep layers: measured 2.0e-6 / 2.8e-6 on configs 1 / 5)
# Operator_1DtMOR, the script's settings, 12 000 epochs (exploration: both tables 0.04-0.17 from epoch 7000 to 60000)
MOR_EPOCHS = 12000
MOR_TABLE_BAR = 0.20        # l2Err(table, field) for kappa = 0.01/pi (cExD3) and kappa = 0.005 (cExD4), all 25 points
MOR_INNER_BAR = 0.08        # the same on the 12 points x <= 0.9, outside the boundary layer (exploration: 0.02-0.07 from epoch 8000)
MOR_CEXACT_BAR = 0.05       # l2Err(cExact(D = 0.1/pi), field at kappa = 0.1/pi) on the script's 100 x 100 grid (exploration: 0.006-0.03)
MOR_FAR_FACTOR = 2.5        # each answer rejects the network evaluated a decade away in kappa by this factor (exploration: 3.5-20)
MOR_SENS_BAR = 0.9          # l2Err((d3 - d4), u(kappa0) - u(kappa1)) on the points x <= 0.9; 1.0 = no kappa-sensitivity at all.  Exploration
                            # (profiles/r4_explore_mor_60000.txt, profiles/r5_explore_mor_120000.txt): 0.79-0.84 at epochs 10 000-14 000,
                            # 0.48-0.69 from epoch 18 000 to 120 000
# BASELINE config 1 = Operator_1Dt at D = 0.1/pi with the 3x20 net, the script's settings, 100 000 epochs
CFG1_EPOCHS = 100000
CFG1_BAR = 0.05             # l2Err(fixData.cEx, evaluate()) (Operator_1Dt.py:177-186); exploration: 0.008-0.028 from epoch 20 000 to 300 000 (uniform throughout, rounds 4-5); round 6, with the script's saveFreq and its one re-draw at epoch 27 800: 0.0062 at epoch 100 000
# Operator_2Dt problem, [40,20] x 40 grid (2.05 M training points), the script's net and weights, 20 000 epochs
OP2_EPOCHS = 20000
OP2_ALL_BAR = 0.20          # the script's metric over ALL 151 time nodes (Operator_2Dt.py:174-183); exploration: plateau 0.133-0.140 on the
                            # [40,20] and on the script's own [80,40] grid alike (the t = 0 node is a discontinuous inlet profile)
OP2_T_BAR = 0.12            # the same at t = T only (exploration: 0.092-0.099)


def tables():
    d = np.load(GOLD)
    return d['inpEx'], d['cEx_kappa_0p01_over_pi'], d['cEx_kappa_0p005'], d['kappa']


def record(key, value):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, 'r4_exact_tables.json')
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


GOLD_CEX = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cexact_1dt.npz')


def test_cexact_restatements_match_reference_outputs():
    """The hand-written restatements of `cExact` used by the tests (tests/test_varnet_host.py) and by the demo
    (examples/operator_1dt.py) against the outputs of the reference's own function (cut out of Operator_1Dt.py:78-108 and
    run, oracle/gen_golden_cexact.py): scattered points incl. the initial line and both boundaries, and the 6 000-point
    grid of the script's acceptance metric.  The MOR script's parametrised copy agrees with it at D = 0.1/pi."""
    import importlib.util
    from tests.test_varnet_host import cExact
    g = np.load(GOLD_CEX)
    np.testing.assert_allclose(g['params'], [1.0, 0.1 / pi, 2.0])
    np.testing.assert_array_equal(g['c'], g['c_mor_D_0p1_over_pi'])
    assert bool(g['mor_refuses_small_D'])                                  # Operator_1DtMOR.py:86-87
    ui = g['uniform_input']
    for name, fn in (('tests', cExact),):
        np.testing.assert_allclose(fn(g['x'].copy(), g['t'].copy()), g['c'], rtol=1e-12, atol=1e-14, err_msg=name)
        np.testing.assert_allclose(fn(ui[:, 0:1].copy(), ui[:, 1:2].copy()), g['c_uniform'], rtol=1e-12, atol=1e-14, err_msg=name)
    spec = importlib.util.spec_from_file_location('ex_operator_1dt', os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'examples', 'operator_1dt.py'))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    np.testing.assert_allclose(ex.cExact(g['x'].copy(), g['t'].copy()), g['c'], rtol=1e-12, atol=1e-14)
    # the fixture's grid is the package's own uniform_input for discNum = 20, tDiscNum = 300 (built there from the formulas)
    vn_ui = _config1_problem(engine=False)
    np.testing.assert_allclose(vn_ui, ui, rtol=0, atol=1e-14)
    on_ic = g['t'][:, 0] == 0
    assert on_ic.sum() == 21 and np.allclose(g['c'][on_ic, 0], -np.sin(pi * g['x'][on_ic, 0]))


def _grad_bar(theta, d_in, widths, kw64, gref, lref=None):
    """Bar for the fp32 gradient at a CONVERGED theta.  Near a minimum the gradient is the small residual of large cancelling
    terms, so its relative fp32 error grows with the cancellation (here 4e-4 where a random theta gives 1e-6).  As in
    tests/fuzz_routes.py the bar follows the measured conditioning of the case, never a global loosening: the stated 1e-4,
    or twice the deviation of the ORACLE's own fp32 run from its fp64 run on the same inputs, whichever is larger."""
    kw32 = {k: (v.astype(np.float32) if isinstance(v, np.ndarray) and v.dtype == np.float64 else v) for k, v in kw64.items()}
    r32, g32 = og.loss_and_grad(theta.astype(np.float32), d_in, widths, torch.float32, **kw32)
    cond = float(np.max(np.abs(np.asarray(g32, dtype=np.float64) - gref)) / np.max(np.abs(gref)))
    if lref is not None:                    # the loss at a converged theta is a sum of squared small residuals R_k: same rule
        lcond = abs(float(r32['loss']) - lref) / abs(lref)
        assert cond <= GRAD_COND_CAP and lcond <= LOSS_COND_CAP, ('theta* is worse conditioned than any converged run so far', cond, lcond)
        return min(max(1e-4, 2.0 * cond), 2.0 * GRAD_COND_CAP), cond, min(max(1e-5, 2.0 * lcond), 2.0 * LOSS_COND_CAP), lcond
    assert cond <= GRAD_COND_CAP, ('theta* is worse conditioned than any converged run so far', cond)
    return min(max(1e-4, 2.0 * cond), 2.0 * GRAD_COND_CAP), cond


# The self-calibrated bars above are CAPPED: the condition estimate itself -- the deviation of the oracle's own fp32 run from
# its fp64 run at theta* -- must stay under these bounds (measured over the three converged runs: gradient 4.1e-4 on config 1,
# loss <= 5e-6), so an ill-conditioned theta* fails the test instead of loosening its bar without limit; the bar never exceeds twice the cap.
# Round 6: config 1 now converges on its RE-DRAWN 144 000-row set (the script's own saveFreq), where the loss is 66 after 1e6 at the start:
# the oracle's own fp32-vs-fp64 deviation there is 4.6e-5 for the loss (5.4e-4 for the gradient).  The loss cap follows that
# measurement with the same 4 x margin the gradient cap has over its largest case.
GRAD_COND_CAP = 1e-3
LOSS_COND_CAP = 2e-4


def _config1_problem(engine=True):
    """BASELINE config 1: Operator_1Dt.py:144-161 (D = 0.1/pi, cEx = cExact) with the 3 x 20 net."""
    from tests.test_varnet_host import cExact
    pde = ADPDE(Domain1D(), diff=0.1 / pi, vel=1.0, timeDependent=True, tInterval=[0, 2.0],
                IC=lambda x: -np.sin(pi * x), cEx=cExact)
    if not engine:                                   # CPU tier: only the discretisation (no engine is created)
        from varnet_amd.varnet import FIXData

        class Shell:
            pass
        sh = VarNet.__new__(VarNet)
        sh.PDE, sh.dim, sh.discNum, sh.bDiscNum, sh.tDiscNum, sh.MORdiscScheme = pde, 1, 20, None, 300, None
        return FIXData(sh).uniform_input
    return VarNet(pde, layerWidth=[20, 20, 20], discNum=20, bDiscNum=None, tDiscNum=300)


@pytest.mark.gpu
def test_config1_converged_run_against_cexact(tmp_path):
    """BASELINE config 1 trained as the script trains it (Operator_1Dt.py:170: smpScheme='optimal', adjustWeight=True, every
    other argument at train()'s default -- saveFreq=100, trainUpdelay=2e4, tolUpd=0.01, reinitrain=True, VarNet.py:1198-1200),
    bounded to 100 000 epochs, then the script's acceptance metric l2Err(fixData.cEx, evaluate()) (:177-186) against the
    reference-generated fixture of its own `cExact`; then the oracle leg at the converged theta*.
    Round 6: through round 5 this test passed saveFreq=10000, and the convergence test that re-draws the
    training set (VarNet.py:1385-1421) looks at the last five losses SAMPLED EVERY saveFreq EPOCHS -- at 10 000 it never fired and
    the "optimal" run was a uniform run.  With the script's own saveFreq the set is re-drawn once (exploration committed before
    this test's first run, profiles/r6_explore_cfg1_optimal_savefreq100.txt: epoch 27 800, then 0.0062 at epoch 100 000): residual-
    driven points ADDED (96 000 -> 144 000 rows), trainable variables re-initialised, BC/IC weights x 5 -- all asserted below."""
    g = np.load(GOLD_CEX)
    vn = _config1_problem()
    fd, eng = vn.fixData, vn.engine
    assert fd.nT == 96000 and eng.P == 921
    np.testing.assert_allclose(fd.cEx, g['c_uniform'], rtol=1e-12, atol=1e-14)      # the metric's cEx IS the reference's output
    np.random.seed(0)
    res = vn.train(str(tmp_path), weight=[10., 10., 1.], smpScheme='optimal', adjustWeight=True, epochNum=CFG1_EPOCHS,
                   verbose=False)
    # the branch of VarNet.py:1385-1421 ran: one re-draw (multiTrainUpd=False) after trainUpdelay, points added, weights x 5
    assert len(res.inpIter) == 1 and res.inpIter[0] >= 20000 - 1, res.inpIter
    assert vn.tData.mor[0]['Input'].shape[0] == 144000 and vn.fixData.nt == 9000          # ceil(0.5 nt) test functions added
    cd = open(os.path.join(str(tmp_path), 'caseData.txt')).read()
    assert 'Training points updated.' in cd and 'trainable variables reinitialized.' in cd
    w_after = np.asarray(res.trainWeight, dtype=float)
    u = vn.evaluate()                                                                # Operator_1Dt.py:179
    e_hip = float(uf.l2Err(g['c_uniform'], u))
    # oracle leg at theta*
    theta = eng.get_params().astype(np.float64)
    u_o = og.forward(theta, 2, [20, 20, 20], torch.float64, fd.uniform_input)
    e_orc = float(uf.l2Err(g['c_uniform'], u_o))
    fdiff = float(np.max(np.abs(u - u_o)))
    from tests.test_operator_parity_gpu import oracle_kwargs
    td = vn.tData                                    # the run's own (re-drawn, 144 000-row) training set: what theta* was trained on
    td.select_mor(0)
    w = np.array([3.0, 2.0, 5.0])
    eng.set_weights(w)
    gb = eng.bind_grad_buffer()
    eng.grad(0)
    torch.cuda.synchronize()
    gh = gb.cpu().numpy().astype(np.float64)
    kw = oracle_kwargs(vn, td, w)
    assert kw['Input'].shape[0] == 144000 and kw['intShape'] == [9000, 16]
    kw = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v) for k, v in kw.items()}
    ref, gref = og.loss_and_grad(theta, 2, [20, 20, 20], torch.float64, **kw)
    lerr = abs(gh[eng.P] - ref['loss']) / abs(ref['loss'])
    gerr = float(np.max(np.abs(gh[:eng.P] - gref)) / np.max(np.abs(gref)))
    # (round 6: the loss at this theta* -- after the re-draw, a sum of 9 000 squared small residuals -- gets the same self-calibrated,
    # capped bar as the gradient and as the other converged runs' losses: 1.02e-5 measured where the oracle's own fp32 run deviates alike)
    gbar, gcond, lbar, lcond = _grad_bar(theta, 2, [20, 20, 20], kw, gref, lref=float(ref['loss']))
    record('config1_converged', dict(epochs=len(res.lossAll), l2Err_cExact_hip=e_hip, l2Err_cExact_oracle_at_theta_star=e_orc,
                                     max_field_diff_hip_vs_oracle=fdiff, loss_rel_err_at_theta_star=float(lerr),
                                     grad_rel_err_at_theta_star=gerr, grad_rel_err_of_the_fp32_restatement_itself=gcond, grad_bar=gbar,
                                     loss_rel_err_of_the_fp32_restatement_itself=lcond, loss_bar=lbar,
                                     loss_first_last=[float(res.lossAll[0]), float(res.lossAll[-1])],
                                     training_sets_redrawn_at=[int(e) for e in res.inpIter], rows_after_the_redraw=144000,
                                     train_weights_at_the_end=[float(x) for x in w_after], bar=CFG1_BAR))
    print('config 1, %d epochs: l2Err(cExact) hip %.5f oracle %.5f, field diff %.1e, loss/grad err at theta* %.1e / %.1e'
          % (len(res.lossAll), e_hip, e_orc, fdiff, lerr, gerr))
    assert e_hip <= CFG1_BAR and e_orc <= CFG1_BAR
    assert fdiff <= FWD_BAR
    assert lerr <= lbar and gerr <= gbar, (lerr, lbar, gerr, gbar)
    eng.close()


# ---------------------------------------------------------------------------------------------------------
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
def test_operator_1dtmor_converged_run_against_the_known_answers(tmp_path):
    """Operator_1DtMOR.py:204 as the script runs it (uniform sampling, 20 shuffled mini-batches per kappa, saveMORdata),
    bounded to 12 000 epochs = 1.44 M Adam steps (about a minute); then Operator_1DtMOR.py:214-229: the two tables, each
    paired with the diffusivity it was tabulated for, and `cExact` at D = 0.1/pi on the script's 100 x 100 grid; the
    discrimination checks of the module docstring; the oracle leg at theta*."""
    inp, d3, d4, kappa = tables()
    g = np.load(GOLD_CEX)
    grid, c_grid = g['mor_input'], g['c_mor_grid_D_0p1_over_pi']
    k_far = 0.1 / pi
    vn = _mor_setup()
    fd, eng = vn.fixData, vn.engine
    np.random.seed(0)
    res = vn.train(str(tmp_path), weight=[10., 10., 1.], smpScheme='uniform', saveMORdata=True, batchNum=20,
                   shuffleData=True, epochNum=MOR_EPOCHS, saveFreq=2000, verbose=False)
    ev = lambda X, k: vn.evaluate(x=X[:, 0:1], t=X[:, 1:2], MORarg=[[k]])          # Operator_1DtMOR.py:217
    theta = eng.get_params().astype(np.float64)
    ev_o = lambda X, k: og.forward(theta, 3, [10, 20, 30], torch.float64, np.hstack([X, k * np.ones([X.shape[0], 1])]))
    inner = inp[:, 0] <= 0.9
    out = {}
    for name, f in (('hip', ev), ('oracle_at_theta_star', ev_o)):
        u0, u1, uf_ = f(inp, kappa[0]), f(inp, kappa[1]), f(inp, k_far)
        out[name] = dict(
            correct=[float(uf.l2Err(d3, u0)), float(uf.l2Err(d4, u1))],
            scripts_swapped_pairing=[float(uf.l2Err(d4, u0)), float(uf.l2Err(d3, u1))],
            inner_points=[float(uf.l2Err(d3[inner], u0[inner])), float(uf.l2Err(d4[inner], u1[inner]))],
            tables_vs_net_a_decade_away=[float(uf.l2Err(d3, uf_)), float(uf.l2Err(d4, uf_))],
            inner_sensitivity=float(uf.l2Err((d3 - d4)[inner], (u0 - u1)[inner])),
            cexact_grid=float(uf.l2Err(c_grid, f(grid, k_far))),
            cexact_grid_vs_net_at_kappa0=float(uf.l2Err(c_grid, f(grid, kappa[0]))))
    fdiff = max(float(np.max(np.abs(ev(inp, k) - ev_o(inp, k)))) for k in (kappa[0], kappa[1], k_far))
    fdiff = max(fdiff, float(np.max(np.abs(ev(grid, k_far) - ev_o(grid, k_far)))))
    # oracle loss / gradient at theta* on one mini-batch of three of the six diffusivity batches (unshuffled feeds)
    _silence(vn)
    td = vn._build_tdata(batchNum=20)
    w = np.array([3.0, 2.0, 5.0])
    eng.set_weights(w)
    gb = eng.bind_grad_buffer()
    q, errs = fd.integNum, []
    f64 = lambda t: t.cpu().numpy().astype(np.float64)
    torch.set_num_threads(16)
    for mb, bi in ((0, 0), (2, 7), (5, 19)):
        td.select_mor(mb)
        eng.grad(td.engine_batch(mb, bi))
        torch.cuda.synchronize()
        gh = gb.cpu().numpy().astype(np.float64)
        d = td.mor[mb]
        n0, n1 = td.block(bi)
        n = (n1 - n0) * q
        kw = dict(Input=f64(d['Input'][n0 * q:n1 * q]), gcoef=f64(d['gcoef'][n0 * q:n1 * q]),
                  source=None, N=np.tile(fd.N, n1 - n0).reshape(n, 1).astype(np.float32).astype(np.float64),
                  dNt=np.tile(fd.dNt, n1 - n0).reshape(n, 1).astype(np.float32).astype(np.float64), integW=None,
                  intShape=[n1 - n0, q], detJ=float(np.float32(fd.detJ)), detJvec=False, biInput=f64(d['biInput']),
                  biLabel=f64(d['biLabel']).reshape(-1, 1), bDof=fd.bDofsum, biDimVal=float(fd.biDimVal), w=w, dim=1,
                  time_dependent=True, is_source=False, integWflag=False)
        ref, gref = og.loss_and_grad(theta, 3, [10, 20, 30], torch.float64, **kw)
        gbar, gcond, lbar, lcond = _grad_bar(theta, 3, [10, 20, 30], kw, gref, ref['loss'])
        errs.append((abs(gh[eng.P] - ref['loss']) / abs(ref['loss']), float(np.max(np.abs(gh[:eng.P] - gref)) / np.max(np.abs(gref))), gbar, gcond,
                     lbar, lcond))
    out.update(epochs=len(res.lossAll), adam_steps=120 * len(res.lossAll), loss_first_last=[float(res.lossAll[0]), float(res.lossAll[-1])],
               max_field_diff_hip_vs_oracle=fdiff,
               loss_grad_rel_err_at_theta_star=[dict(loss=float(a), grad=float(b), grad_bar=float(c), grad_err_of_the_fp32_restatement_itself=float(e),
                                                     loss_bar=float(lb), loss_err_of_the_fp32_restatement_itself=float(lc))
                                                for a, b, c, e, lb, lc in errs],
               tables_relative_distance=float(uf.l2Err(d3, d4)),
               bars=dict(table=MOR_TABLE_BAR, inner=MOR_INNER_BAR, cexact=MOR_CEXACT_BAR, far_factor=MOR_FAR_FACTOR, inner_sensitivity=MOR_SENS_BAR))
    record('operator_1dtmor_converged', out)
    print('MOR converged run: %s' % json.dumps(out['hip']))
    for name in ('hip', 'oracle_at_theta_star'):
        o = out[name]
        assert max(o['correct']) <= MOR_TABLE_BAR, (name, o)
        assert max(o['inner_points']) <= MOR_INNER_BAR, (name, o)
        assert o['cexact_grid'] <= MOR_CEXACT_BAR, (name, o)
        # discrimination: every answer rejects the network evaluated a decade away in kappa ...
        assert o['tables_vs_net_a_decade_away'][0] >= MOR_FAR_FACTOR * o['correct'][0], (name, o)
        assert o['tables_vs_net_a_decade_away'][1] >= MOR_FAR_FACTOR * o['correct'][1], (name, o)
        assert o['cexact_grid_vs_net_at_kappa0'] >= MOR_FAR_FACTOR * o['cexact_grid'], (name, o)
        # ... and outside the boundary layer the kappa-sensitivity follows the tables' difference (1.0 = no sensitivity at all)
        assert o['inner_sensitivity'] <= MOR_SENS_BAR, (name, o)
    assert fdiff <= FWD_BAR
    assert all(e[0] <= e[4] and e[1] <= e[2] for e in errs), errs
    eng.close()


# ---------------------------------------------------------------------------------------------------------
# Operator_2Dt.py:89-132: the reference's analytical solution of its 2D+t demo (Leij & Dane, integrated over time), the
# known answer behind BASELINE config 3's problem.  tests/golden/cexfun_2dt.npz = outputs of the reference's own function
# (oracle/gen_golden_cexfun.py); the metric is the script's: l2Err over ALL 151 time nodes (Operator_2Dt.py:174-183).
GOLD2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cexfun_2dt.npz')


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
def test_operator_2dt_budget_run_hip_vs_oracle():
    from tests.test_operator_parity_gpu import run_both
    from tests.test_varnet_gpu import op2dt
    g = np.load(GOLD2)
    vn = op2dt([10, 20], [20, 10], 10, 15)                             # Operator_2Dt.py:136-158, grid scaled down
    gl, cl, th_g, th_c, _ = run_both(vn, [5., 1., 1.], 400)          # weight: Operator_2Dt.py:167
    e_g = _l2err_2dt(lambda X: vn.evaluate(x=X[:, :2], t=X[:, 2:3]), g)
    e_c = _l2err_2dt(lambda X: og.forward(th_c.astype(np.float64), 3, [10, 20], torch.float64, X), g)
    dev = float(np.max(np.abs(gl - cl) / np.abs(cl)))
    record('operator_2dt_budget', dict(budget_steps=400, max_rel_loss_dev=dev, l2Err_cExFun_hip=e_g, l2Err_cExFun_oracle=e_c))
    assert dev <= 1e-2 and abs(e_g - e_c) <= 1e-4, (dev, e_g, e_c)
    vn.engine.close()


@pytest.mark.gpu
def test_operator_2dt_converged_run_against_the_analytical_solution(tmp_path):
    """Operator_2Dt.py:136-167 (the script's net [10,20] and weights [5,1,1]) on a [40,20] x 40 grid = 2.05 M training points,
    20 000 epochs; then the script's metric (:174-183) over all 151 time nodes against the reference-generated outputs of
    `cExFun`, the same at t = T, and the oracle's own distance at theta*."""
    from tests.test_varnet_gpu import op2dt
    g = np.load(GOLD2)
    vn = op2dt([10, 20], [40, 20], 20, 40)
    eng = vn.engine
    np.random.seed(0)
    res = vn.train(str(tmp_path), weight=[5., 1., 1.], smpScheme='uniform', epochNum=OP2_EPOCHS, saveFreq=4000, verbose=False, lossLag=16)
    x, T = g['x'], float(g['params'][0])
    theta = eng.get_params().astype(np.float64)
    fwd = {'hip': lambda X: vn.evaluate(x=X[:, :2], t=X[:, 2:3]),
           'oracle_at_theta_star': lambda X: og.forward(theta, 3, [10, 20], torch.float64, X)}
    out = {}
    XT = np.hstack([x, T * np.ones([len(x), 1])])
    for name, f in fwd.items():
        out[name] = dict(all_time_nodes=_l2err_2dt(f, g), at_T=float(uf.l2Err(g['c_all'][:, -1:], f(XT))))
    fdiff = float(np.max(np.abs(fwd['hip'](XT) - fwd['oracle_at_theta_star'](XT))))
    out.update(epochs=len(res.lossAll), training_points=int(vn.fixData.nT), loss_first_last=[float(res.lossAll[0]), float(res.lossAll[-1])],
               max_field_diff_hip_vs_oracle=fdiff, bars=dict(all_time_nodes=OP2_ALL_BAR, at_T=OP2_T_BAR))
    record('operator_2dt_converged', out)
    print('2Dt converged run: %s' % json.dumps(out))
    for name in fwd:
        assert out[name]['all_time_nodes'] <= OP2_ALL_BAR and out[name]['at_T'] <= OP2_T_BAR, (name, out[name])
    assert fdiff <= FWD_BAR
    eng.close()
