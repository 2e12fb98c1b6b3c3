"""
Networks outside the range of the fused / generic kernels (more than 6 hidden layers, widths above 64, more than 8
inputs) -- the reference takes any `layerWidth` (TFModel.py:208-221).  They run on the layer-by-layer route
(vn_layered.hip); same oracle, same bars as tests/test_engine_gpu.py.  The route is also forced (kernel=4) on networks
the kernels cover, where it must agree with them.
"""
import numpy as np
import pytest
import torch

from oracle import tf1_graph as og
from tests.test_engine_gpu import synth, make_engine, oracle_eval, LOSS_RTOL, GRAD_RTOL, LVEC_RTOL

pytestmark = pytest.mark.gpu

LAYERED = 4

CASES = [
    # d_in dim widths                 integNum n_k  nB  bDof source integW detJvec
    (3, 2, [100, 80],                 64,      9,   77, 40,  False, False, False),   # wider than 64
    (3, 2, [128, 128, 128],           64,      30,  50, 20,  True,  False, True),
    (3, 2, [96, 80, 96, 72, 96],      16,      21,  40, 15,  False, True,  False),   # tile kernels: <6,2,3> instantiation
    (3, 2, [128, 112, 128, 100, 128, 120], 16, 9,   30, 12,  True,  False, False),   # ... layer-serial reverse (5+ layers wider than 96)
    (2, 1, [64, 50, 64, 33, 64, 64, 40, 64, 64], 16, 15, 30, 10, False, False, True),   # ... <16,1,2>
    (3, 2, [200, 256, 130, 180],      16,      7,   20, 9,   True,  True,  False),   # ... two row-tile passes, layer-serial reverse
    (4, 3, [128] * 8,                 8,       9,   20, 9,   False, False, False),   # ... layer-serial reverse at 128
    (2, 1, [20] * 8,                  16,      40,  50, 30,  False, False, False),   # more than 6 hidden layers
    (3, 2, [30, 70, 12, 65, 9, 40, 33], 36,    17,  12, 5,   True,  True,  True),    # ragged, 7 layers
    (10, 2, [40, 40],                 64,      12,  30, 10,  False, True,  False),   # more than 8 inputs (many MOR parameters)
    (3, 2, [256],                     216,     3,   5,  2,   False, True,  False),   # integNum 216, one wide layer
    (3, 2, [50, 50, 50],              64,      40,  30, 10,  True,  False, False),   # kernel range: forced route
    (2, 1, [7],                       36,      11,  40, 13,  True,  True,  True),
    (3, 2, [2048, 700],               16,      5,   9,  4,   False, False, False),   # the widest a vn_config describes
    (2, 1, [12] * 16,                 16,      20,  30, 10,  True,  False, False),   # ... and the deepest
    (32, 3, [24, 24],                 64,      6,   20, 8,   False, True,  False),   # ... and the most inputs
]


def _setup(case, kernel, act='sigmoid'):
    d_in, dim, widths, integNum, n_k, nB, bDof, source, integW, detJvec = case
    d = synth(1, d_in, dim, widths, integNum, n_k, nB, bDof, source, integW, detJvec)
    from varnet_amd.engine import VNEngine
    eng = VNEngine(dim, d_in, widths, True, integNum, isSource=source, integWflag=integW, kernel=kernel, activationFun=act)
    eng.init_params(seed=3)
    flat = eng.get_params()
    flat = flat + 0.05 * np.random.default_rng(5).standard_normal(flat.size).astype(np.float32)
    eng.set_params(flat)
    eng.set_fe_table(d['N1'], d['dNt1'], d['integW'])
    eng.set_interior(0, d['Input'], d['gcoef'], d['source'], n_k=n_k, detJ=d['detJ'])
    eng.set_bic(d['biInput'], d['biLabel'], bDof, 2.0)
    eng.set_weights(d['w'])
    return eng, d, flat


def _tile_kernels_take(widths, d_in):
    """vn_wide_supported (vn_wide.hip): hidden widths up to 256;
    at most 32 inputs"""
    return d_in <= 32 and (max(widths) <= 256)


@pytest.mark.parametrize('impl', ['tile-kernels', 'gemms'])
@pytest.mark.parametrize('case', CASES)
def test_loss_and_grad_parity_layered(case, impl, monkeypatch):
    """Both implementations of the route against the fp64 oracle: the tile kernels of vn_wide.hip (nets up to 128 wide) and the
    GEMM form (every net; VN_LAYERED_NOWIDE=1 keeps a net the tile kernels would take on it)."""
    d_in, dim, widths, integNum, n_k, nB, bDof, source, integW, detJvec = case
    if impl == 'gemms':
        monkeypatch.setenv('VN_LAYERED_NOWIDE', '1')
    elif not _tile_kernels_take(widths, d_in):
        pytest.skip('beyond the tile kernels: covered by the gemms variant')
    eng, d, flat = _setup(case, LAYERED)
    assert eng.kernel_path()[0] == LAYERED
    if len(widths) > 8 or max(widths) > 64 or d_in > 8 or (len(widths) > 6 and max(widths) > 50):
        auto = _setup(case, 0)[0]                   # AUTO resolves to this route for nets beyond the kernels
        assert auto.kernel_path()[0] == LAYERED
        auto.close()
    ref, gref = oracle_eval(flat, d, d_in, dim, widths, integNum, n_k, bDof, source, integW, detJvec)
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
    err = np.max(np.abs(g[:eng.P] - gref)) / np.max(np.abs(gref))
    assert err <= GRAD_RTOL, err
    # run-to-run reproducible (no atomics in the weight-gradient GEMMs)
    eng.grad(0)
    torch.cuda.synchronize()
    assert np.array_equal(gb.cpu().numpy(), g)
    eng.close()


def test_layered_tanh_and_agreement_with_the_kernels():
    case = (3, 2, [50, 50, 50], 64, 40, 30, 10, True, False, False)
    grads = []
    for kernel in (0, LAYERED):
        eng, d, flat = _setup(case, kernel, act='tanh')
        gb = eng.bind_grad_buffer()
        eng.grad(0)
        torch.cuda.synchronize()
        grads.append(gb.cpu().numpy().astype(np.float64))
        eng.close()
    a, b = grads
    assert np.max(np.abs(a[:-4] - b[:-4])) <= 3e-5 * np.max(np.abs(a[:-4]))
    assert abs(a[-4] - b[-4]) <= 1e-5 * abs(a[-4])


@pytest.mark.parametrize('keep', [True, False], ids=['kept-activations', 'recompute'])
@pytest.mark.parametrize('widths', [[96, 96, 96], [160, 144], [272, 260]], ids=['tile-kernels', 'tile-kernels-layer-serial', 'gemms'])
def test_layered_chunks_and_shard_additivity(widths, keep, monkeypatch):
    """Both forms of the reverse pass: reading the activations the forward kept in HBM, and recomputing them per chunk
    (VN_LAYERED_NOKEEP=1, what happens when they do not fit).  1.28 M rows do not fit the route's workspace in one piece: the
    interior set is processed in several chunks, and the same set fed as two halves (chunked differently) sums to the same
    gradient and loss.  Up to 256 wide the kept form runs on the tile kernels of vn_wide.hip and the recompute form on
    the GEMMs (two implementations: agreement to fp32 rounding); beyond, both forms are the GEMMs and give the same bits."""
    d_in, dim, integNum, n_k, nB, bDof = 3, 2, 64, 20000, 3000, 1700
    d = synth(7, d_in, dim, widths, integNum, n_k, nB, bDof)
    if not keep:
        monkeypatch.setenv('VN_LAYERED_NOKEEP', '1')
    from varnet_amd.engine import VNEngine
    eng = VNEngine(dim, d_in, widths, True, integNum, kernel=0)
    assert eng.kernel_path()[0] == LAYERED
    eng.init_params(seed=2)
    eng.set_fe_table(d['N1'], d['dNt1'], None)
    eng.set_interior(0, d['Input'], d['gcoef'], None, n_k=n_k, detJ=d['detJ'])
    eng.set_bic(d['biInput'], d['biLabel'], bDof, 2.0)
    eng.set_weights(d['w'])
    gb = eng.bind_grad_buffer()
    eng.grad(0)
    torch.cuda.synchronize()
    g = gb.cpu().numpy().astype(np.float64)
    # shard additivity: the same set as two halves (different chunking) sums to the whole
    h = n_k // 2 * integNum
    parts = []
    for sl, k in ((slice(0, h), n_k // 2), (slice(h, None), n_k - n_k // 2)):
        eng.set_interior(0, d['Input'][sl], d['gcoef'][sl], None, n_k=k, detJ=d['detJ'])
        eng.set_weights(d['w'] * np.array([0.5, 0.5, 1.0]))        # BC/IC replicated on both halves at half weight
        eng.grad(0)
        torch.cuda.synchronize()
        parts.append(gb.cpu().numpy().astype(np.float64))
    s = parts[0] + parts[1]
    assert np.max(np.abs(s[:eng.P] - g[:eng.P])) <= 2e-5 * np.max(np.abs(g[:eng.P]))
    assert abs(s[eng.P] - g[eng.P]) <= 1e-5 * abs(g[eng.P])
    eng.close()
    # the two forms of the reverse pass give the same bits (same chunks, same kernels, same order)
    key = 'layered_chunks_grad_%d' % widths[0]
    prev = getattr(test_layered_chunks_and_shard_additivity, key, None)
    if prev is not None:
        if max(widths) > 256:
            assert np.array_equal(prev, g)
        else:
            assert np.max(np.abs(prev[:-4] - g[:-4])) <= 3e-5 * np.max(np.abs(g[:-4]))
            assert abs(prev[-4] - g[-4]) <= 1e-5 * abs(g[-4])
    setattr(test_layered_chunks_and_shard_additivity, key, g)


def _gemm_symbols():
    """The hand-written products of vn_gemm.hip as libvarnet_hip.so exports them (C++ linkage: vn_internal.h:180-192); the
    library form they are checked against is torch.matmul on the same device, i.e. the vendor GEMMs behind PyTorch-ROCm.
    The product library itself neither links nor loads a vendor library (round 4: the VN_LAYERED_ROCBLAS twin is gone)."""
    import ctypes as C
    from varnet_amd.engine import load_library
    lib = load_library()
    vp, i, l = C.c_void_p, C.c_int, C.c_long
    sig = {'nn': ('_Z10vn_gemm_nnPKfS0_PfliiP12ihipStream_t', [vp, vp, vp, l, i, i, vp]),
           'fwd': ('_Z11vn_gemm_fwdPKfS0_S0_PfliiiiP12ihipStream_t', [vp, vp, vp, vp, l, i, i, i, i, vp]),
           'tn_rows': ('_Z15vn_gemm_tn_rowsliii', [l, i, i, i]),
           'tn_parts': ('_Z16vn_gemm_tn_partsPKfS0_PfliilP12ihipStream_t', [vp, vp, vp, l, i, i, l, vp]),
           'rowdot': ('_Z9vn_rowdotPKfS0_PflifP12ihipStream_t', [vp, vp, vp, l, i, C.c_float, vp]),
           'dnn': ('_Z11vn_dgemm_nnPKdS0_PdliiP12ihipStream_t', [vp, vp, vp, l, i, i, vp]),
           'drowdot': ('_Z10vn_drowdotPKdS0_PdlidP12ihipStream_t', [vp, vp, vp, l, i, C.c_double, vp])}
    out = {}
    for k, (name, args) in sig.items():
        f = getattr(lib, name)
        f.argtypes, f.restype = args, (C.c_long if k == 'tn_rows' else C.c_int)
        out[k] = f
    return out


@pytest.mark.parametrize('K,N', [(272, 260), (300, 262), (262, 257), (512, 384), (3, 300), (700, 2048)],
                         ids=['x4-widths', 'odd-widths', 'odd-odd', 'tile-multiples', 'input-layer', 'widest'])
def test_hand_written_gemms_against_the_library_form(K, N):
    """Every product of the GEMM route (vn_gemm.hip: 128 x 128 / 128 x 256 C tiles, thin input-layer kernels, fused forward
    epilogue, row-group partial products of the weight gradient, fp64 MFMA product) against the vendor library's GEMMs on the
    same operands: vectorised and element-wise load paths, ragged tiles in all three dimensions, several row groups.
    Two implementations of one product must agree to fp32 rounding of a K- (or M-) term sum; the hand-written form must be
    run-to-run reproducible."""
    g = _gemm_symbols()
    dev = torch.device('cuda')
    gen = torch.Generator(device='cuda').manual_seed(K * 4099 + N)
    S, c = 2, 96133                                  # stacked (value | tangent) rows, not a multiple of any tile
    M = S * c
    A = torch.rand(M, K, device=dev, generator=gen) - 0.3
    W = (torch.rand(K, N, device=dev, generator=gen) - 0.5) * (2.0 / K ** 0.5)
    bias = torch.rand(N, device=dev, generator=gen) - 0.5
    st = torch.cuda.current_stream().cuda_stream
    tol = lambda ref, k: 4e-7 * k ** 0.5 * float(ref.abs().max()) + 1e-7
    # plain product
    C1 = torch.empty(M, N, device=dev)
    assert g['nn'](A.data_ptr(), W.data_ptr(), C1.data_ptr(), M, N, K, st) == 0
    ref = A @ W
    assert float((C1 - ref).abs().max()) <= tol(ref, K)
    C2 = torch.empty_like(C1)
    assert g['nn'](A.data_ptr(), W.data_ptr(), C2.data_ptr(), M, N, K, st) == 0
    assert torch.equal(C1, C2)
    # forward with the layer epilogue: (sigmoid(A0 W + b) | sigmoid'(.) (A1 W)); act code 0 = sigmoid (vn_internal.h)
    assert g['fwd'](A.data_ptr(), W.data_ptr(), bias.data_ptr(), C1.data_ptr(), c, S, N, K, 0, st) == 0
    a = torch.sigmoid(ref[:c] + bias)
    ad = a * (1 - a) * ref[c:]
    assert float((C1[:c] - a).abs().max()) <= 2e-6 and float((C1[c:] - ad).abs().max()) <= tol(ref, K)
    # weight gradient: A^T Z in row groups, summed in fixed order
    Z = torch.rand(M, N, device=dev, generator=gen) - 0.5
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    rows = g['tn_rows'](M, K, N, ncu)
    groups = (M + rows - 1) // rows
    assert 1 <= rows <= M and groups >= 2            # several row groups, the last one ragged
    parts = torch.empty(groups, K, N, device=dev)
    assert g['tn_parts'](A.data_ptr(), Z.data_ptr(), parts.data_ptr(), M, K, N, rows, st) == 0
    refT = (A.double().T @ Z.double())
    got = parts.double().sum(0)
    assert float((got - refT).abs().max()) <= 4e-7 * rows ** 0.5 * float(refT.abs().max()) * groups ** 0.5
    lib32 = A.T @ Z                                   # the library's own fp32 result is no closer to fp64 than ours by more than 4x
    assert float((got - refT).abs().max()) <= 4 * float((lib32.double() - refT).abs().max()) + 1e-6 * float(refT.abs().max())
    # output layer: y = beta y + A w
    w = torch.rand(K, device=dev, generator=gen) - 0.5
    y = torch.ones(M, device=dev)
    assert g['rowdot'](A.data_ptr(), w.data_ptr(), y.data_ptr(), M, K, 0.5, st) == 0
    refy = 0.5 + A @ w
    assert float((y - refy).abs().max()) <= tol(refy, K)
    # fp64 MFMA product and output layer of the fp64 entry points
    Md = 4099
    Ad, Wd = A[:Md].double(), W.double()
    Cd = torch.empty(Md, N, device=dev, dtype=torch.float64)
    assert g['dnn'](Ad.data_ptr(), Wd.data_ptr(), Cd.data_ptr(), Md, N, K, st) == 0
    refd = Ad @ Wd
    assert float((Cd - refd).abs().max()) <= 1e-13 * K ** 0.5 * max(1.0, float(refd.abs().max()))
    yd = torch.zeros(Md, device=dev, dtype=torch.float64)
    assert g['drowdot'](Ad.data_ptr(), w.double().data_ptr(), yd.data_ptr(), Md, K, 0.0, st) == 0
    assert float((yd - Ad @ w.double()).abs().max()) <= 1e-13 * K ** 0.5
    torch.cuda.synchronize()


def test_gemm_route_multi_chunk_multi_tile_against_the_oracle():
    """the > 256-wide route at scale was only checked peer-vs-peer.  [300, 262] net, 3 001 test functions x
    64 points = 192 064 rows x 2 streams (several chunks of the 1 GB workspace, several 128 x 256 C tiles per product, several
    row groups per weight gradient) against the fp64 oracle evaluated block by block: the variational term is a sum over
    test functions (TFModel.py:655-668), so blocks of 300 test functions with weights (0, 0, w_var) add up, and the BC/IC
    terms come from one more call with (w_bc, w_ic, 0)."""
    d_in, dim, widths, integNum, n_k, nB, bDof = 3, 2, [300, 262], 64, 3001, 1501, 700
    d = synth(9, d_in, dim, widths, integNum, n_k, nB, bDof)
    from varnet_amd.engine import VNEngine
    eng = VNEngine(dim, d_in, widths, True, integNum, kernel=0)
    assert eng.kernel_path()[0] == LAYERED
    eng.init_params(seed=4)
    eng.set_fe_table(d['N1'], d['dNt1'], None)
    eng.set_interior(0, d['Input'], d['gcoef'], None, n_k=n_k, detJ=d['detJ'])
    eng.set_bic(d['biInput'], d['biLabel'], bDof, 2.0)
    eng.set_weights(d['w'])
    gb = eng.bind_grad_buffer()
    eng.grad(0)
    torch.cuda.synchronize()
    g = gb.cpu().numpy().astype(np.float64)
    eng.grad(0)
    torch.cuda.synchronize()
    assert np.array_equal(gb.cpu().numpy().astype(np.float64), g)              # run-to-run reproducible
    flat = eng.get_params().astype(np.float64)
    P = eng.P
    torch.set_num_threads(16)
    w = np.asarray(d['w'], dtype=np.float64)
    loss, grad = 0.0, np.zeros(P)
    blk = 300
    for k0 in range(0, n_k, blk):
        k1 = min(n_k, k0 + blk)
        sub = dict(d)
        r0, r1 = k0 * integNum, k1 * integNum
        sub['Input'], sub['gcoef'], sub['N'], sub['dNt'] = d['Input'][r0:r1], d['gcoef'][r0:r1], d['N'][r0:r1], d['dNt'][r0:r1]
        sub['w'] = np.array([w[0], w[1], w[2]]) if k0 == 0 else np.array([0.0, 0.0, w[2]])
        ref, gref = oracle_eval(flat, sub, d_in, dim, widths, integNum, k1 - k0, bDof, False, False, False)
        loss += ref['loss']
        grad += gref
    assert abs(g[P] - loss) <= LOSS_RTOL * abs(loss), (g[P], loss)
    err = np.max(np.abs(g[:P] - grad)) / np.max(np.abs(grad))
    assert err <= GRAD_RTOL, err
    u = eng.forward(d['Input'][:4099]).cpu().numpy()
    uref = og.forward(flat, d_in, widths, torch.float64, np.asarray(d['Input'][:4099], dtype=np.float64))
    assert np.max(np.abs(u - uref[:, 0])) <= 2e-6 * max(1.0, np.max(np.abs(uref)))
    eng.close()


@pytest.mark.parametrize('widths,d_in,dim', [([100, 80], 3, 2), ([20] * 9, 2, 1), ([40, 40], 10, 3)])
def test_forward_and_residual_parity_layered(widths, d_in, dim):
    rng = np.random.default_rng(0)
    n = 1000
    X = rng.uniform(-1, 1, (n, d_in))
    diff = rng.uniform(0.1, 1, (n, 1)); vel = rng.standard_normal((n, dim))
    src = rng.standard_normal((n, 1)); ddx = rng.standard_normal((n, dim))
    eng = make_engine(d_in, dim, widths, 64, False, False)
    assert eng.kernel_path()[0] == LAYERED
    eng.init_params(seed=11)
    flat = eng.get_params().astype(np.float64)
    uref, rref = og.residual(flat, d_in, widths, torch.float64, X, diff, vel, src, ddx, dim, True)
    u32 = eng.forward(X.astype(np.float32)).cpu().numpy()
    assert np.max(np.abs(u32 - uref[:, 0])) < 2e-6 * max(1, np.max(np.abs(uref)))
    u64 = eng.forward_f64(X).cpu().numpy()
    assert np.max(np.abs(u64 - uref[:, 0])) < 1e-13
    u, r = eng.residual(X, diff, vel, src, ddx, fp64=True)
    assert np.max(np.abs(r.cpu().numpy() - rref[:, 0])) < 1e-11 * max(1, np.max(np.abs(rref)))
    assert np.max(np.abs(u.cpu().numpy() - uref[:, 0])) < 1e-13
    u, r = eng.residual(X.astype(np.float32), diff, vel, src, ddx, fp64=False)
    assert np.max(np.abs(r.cpu().numpy() - rref[:, 0])) < 5e-5 * max(1, np.max(np.abs(rref)))
    eng.close()


def test_adam_trajectory_parity_layered():
    """100 TF-1 Adam steps of a 2 x 100 net from identical init against the fp64 oracle (<= 1e-2)."""
    d_in, dim, widths, integNum, n_k, nB, bDof = 2, 1, [100, 100], 16, 64, 60, 40
    d = synth(2, d_in, dim, widths, integNum, n_k, nB, bDof)
    eng = make_engine(d_in, dim, widths, integNum, False, False, 0)
    eng.init_params(seed=1)
    flat = eng.get_params()
    eng.set_fe_table(d['N1'], d['dNt1'], None)
    eng.set_interior(0, d['Input'], d['gcoef'], None, n_k=n_k, detJ=d['detJ'])
    eng.set_bic(d['biInput'], d['biLabel'], bDof, 2.0)
    eng.set_weights(d['w'])
    steps = 100
    losses = torch.zeros(steps, device='cuda')
    for i in range(steps):
        eng.train_step(0, losses[i:i + 1])
    torch.cuda.synchronize()
    got = losses.cpu().numpy()
    adam = og.TF1Adam(flat.size, lr=1e-3, dtype=np.float64)
    th = flat.astype(np.float64)
    ref = []
    for i in range(steps):
        r, g = oracle_eval(th, d, d_in, dim, widths, integNum, n_k, bDof, False, False, False)
        ref.append(r['loss'])
        th = adam.step(th, g)
    ref = np.array(ref)
    assert np.max(np.abs(got - ref) / np.abs(ref)) <= 1e-2
    assert np.max(np.abs(eng.get_params() - th)) <= 2e-3 * np.max(np.abs(th))
    eng.close()


@pytest.mark.parametrize('widths', [[128, 96, 128], [128] * 7, [200, 256]], ids=['one-launch', 'layer-serial', 'two-passes'])
def test_long_trajectory_tile_kernels_against_the_gemm_form(widths, monkeypatch):
    """600 TF-1 Adam steps from the same start on both implementations of the route (tile kernels of vn_wide.hip, GEMM form): two
    fp32 programs with different summation orders stay within 1e-3 in relative loss, and end at the same parameters to 1e-3."""
    d_in, dim, integNum, n_k, nB, bDof = 3, 2, 64, 37, 90, 50
    d = synth(11, d_in, dim, widths, integNum, n_k, nB, bDof)
    runs = []
    for gemms in (False, True):
        if gemms:
            monkeypatch.setenv('VN_LAYERED_NOWIDE', '1')
        eng = make_engine(d_in, dim, widths, integNum, False, False, 0)
        monkeypatch.delenv('VN_LAYERED_NOWIDE', raising=False)
        assert eng.kernel_path()[0] == LAYERED
        eng.init_params(seed=4)
        eng.set_fe_table(d['N1'], d['dNt1'], None)
        eng.set_interior(0, d['Input'], d['gcoef'], None, n_k=n_k, detJ=d['detJ'])
        eng.set_bic(d['biInput'], d['biLabel'], bDof, 2.0)
        eng.set_weights(d['w'])
        steps = 600
        losses = torch.zeros(steps, device='cuda')
        for i in range(steps):
            eng.train_step(0, losses[i:i + 1])
        torch.cuda.synchronize()
        runs.append((losses.cpu().numpy().astype(np.float64), eng.get_params().astype(np.float64)))
        eng.close()
    (la, ta), (lb, tb) = runs
    assert la[-1] < la[0]
    assert np.max(np.abs(la - lb) / np.abs(lb)) <= 1e-3
    assert np.max(np.abs(ta - tb)) <= 1e-3 * np.max(np.abs(tb))


def test_varnet_constructor_accepts_a_wide_deep_net(tmp_path):
    """The kept constructor with a layerWidth the kernels do not cover trains end to end (Operator_1Dt problem)."""
    from tests.test_varnet_host import op1dt
    vn = op1dt(layerWidth=[80, 80, 80, 80, 80, 80, 80], discNum=10, tDiscNum=20)
    res = vn.train(str(tmp_path), weight=[10., 10., 1.], epochNum=30, saveFreq=10, verbose=False)
    assert vn.engine.kernel_path()[0] == LAYERED
    assert np.isfinite(res.lossAll).all() and res.lossAll[-1] < res.lossAll[0]
    r64, _, _, _ = vn.residual(fp64=True)
    r32, _, _, _ = vn.residual()
    assert np.isfinite(r64) and abs(r64 - r32) <= 1e-3 * abs(r64)


@pytest.mark.parametrize('widths,acts', [([20, 30, 20], ['tanh', 'sigmoid', 'tanh']), ([100, 80], ['sigmoid', 'tanh'])])
def test_per_layer_activation_lists(widths, acts):
    """activationFun as one entry per hidden layer (TFModel.py:113-119): entries that differ run layer by layer;
    loss, gradient, forward and fp64 residual against the oracle with the same list."""
    d_in, dim, q, n_k, nB, bDof = 3, 2, 64, 21, 30, 12
    d = synth(3, d_in, dim, widths, q, n_k, nB, bDof, True, False, False)
    from varnet_amd.engine import VNEngine
    eng = VNEngine(dim, d_in, widths, True, q, isSource=True, activationFun=acts)
    assert eng.kernel_path()[0] == LAYERED
    eng.init_params(seed=3)
    flat = eng.get_params() + 0.05 * np.random.default_rng(5).standard_normal(eng.P).astype(np.float32)
    eng.set_params(flat)
    eng.set_fe_table(d['N1'], d['dNt1'], None)
    eng.set_interior(0, d['Input'], d['gcoef'], d['source'], n_k=n_k, detJ=d['detJ'])
    eng.set_bic(d['biInput'], d['biLabel'], bDof, 2.0)
    eng.set_weights(d['w'])
    f64 = lambda a: None if a is None else a.astype(np.float64)
    ref, gref = og.loss_and_grad(
        flat.astype(np.float64), d_in, widths, torch.float64, Input=f64(d['Input']), gcoef=f64(d['gcoef']),
        source=f64(d['source']), N=f64(d['N']), dNt=f64(d['dNt']), integW=None, intShape=[n_k, q],
        detJ=float(d['detJ']), detJvec=False, biInput=f64(d['biInput']), biLabel=f64(d['biLabel']), bDof=bDof,
        biDimVal=2.0, w=d['w'], dim=dim, time_dependent=True, is_source=True, integWflag=False, activation=acts)
    gb = eng.bind_grad_buffer()
    eng.grad(0)
    torch.cuda.synchronize()
    g = gb.cpu().numpy()
    assert abs(g[eng.P] - ref['loss']) <= LOSS_RTOL * abs(ref['loss'])
    assert np.max(np.abs(g[:eng.P] - gref)) / np.max(np.abs(gref)) <= GRAD_RTOL
    rng = np.random.default_rng(0)
    n = 500
    X = rng.uniform(-1, 1, (n, d_in))
    diff = rng.uniform(0.1, 1, (n, 1)); vel = rng.standard_normal((n, dim))
    src = rng.standard_normal((n, 1)); ddx = rng.standard_normal((n, dim))
    uref, rref = og.residual(flat.astype(np.float64), d_in, widths, torch.float64, X, diff, vel, src, ddx, dim, True,
                             activation=acts)
    u, r = eng.residual(X, diff, vel, src, ddx, fp64=True)
    assert np.max(np.abs(r.cpu().numpy() - rref[:, 0])) < 1e-11 * max(1, np.max(np.abs(rref)))
    assert np.max(np.abs(u.cpu().numpy() - uref[:, 0])) < 1e-13
    eng.close()
    # a list whose entries agree is the uniform case and stays on the kernels
    eng = VNEngine(dim, d_in, [20, 30, 20], True, q, activationFun=['tanh'] * 3)
    assert eng.kernel_path()[0] == 3
    eng.close()
    eng = VNEngine(dim, d_in, [20, 30, 20], True, q, activationFun=['tanh'])
    assert eng.kernel_path()[0] == 3
    eng.close()
    with pytest.raises(ValueError):
        VNEngine(dim, d_in, [20, 30, 20], True, q, activationFun=['tanh', 'sigmoid'])     # length != depth


DEEP = [
    # 7 and 8 hidden layers up to 50 wide are instantiated in the 8-wave fused kernel (no generic kernels for them)
    (2, 1, [20] * 8,                     16,  40, 50, 30, False, False, False),
    (3, 2, [32] * 7,                     64,  30, 50, 20, True,  False, True),
    (3, 2, [32] * 8,                     36,  17, 12, 5,  True,  True,  False),
    (3, 2, [20, 10, 20, 7, 20, 13, 20],  216, 5,  9,  4,  False, True,  False),   # two-pass route
    (3, 2, [50] * 7,                     64,  30, 50, 20, False, False, False),   # 50 wide: heavy register spilling, still the kernel
    (3, 2, [50, 33, 50, 40, 50, 21, 50, 50], 36, 17, 12, 5, True, True, True),
]


@pytest.mark.parametrize('act', ['sigmoid', 'tanh'])
@pytest.mark.parametrize('case', DEEP)
def test_deep_narrow_nets_on_the_fused_kernel(case, act):
    d_in, dim, widths, integNum, n_k, nB, bDof, source, integW, detJvec = case
    eng, d, flat = _setup(case, 0, act=act)
    assert eng.kernel_path()[0] == 3                      # 8-wave fused kernel (two-pass for integNum 216)
    f64 = lambda a: None if a is None else a.astype(np.float64)
    ref, gref = og.loss_and_grad(
        flat.astype(np.float64), d_in, widths, torch.float64, Input=f64(d['Input']), gcoef=f64(d['gcoef']),
        source=f64(d['source']), N=f64(d['N']), dNt=f64(d['dNt']), integW=f64(d['integW']), intShape=[n_k, integNum],
        detJ=(f64(d['detJ']) if detJvec else float(d['detJ'])), detJvec=detJvec, biInput=f64(d['biInput']),
        biLabel=f64(d['biLabel']), bDof=bDof, biDimVal=2.0, w=d['w'], dim=dim, time_dependent=True,
        is_source=source, integWflag=integW, activation=act)
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
    # forward / residual entry points of such an engine (pointwise kernels, fused forward)
    rng = np.random.default_rng(0)
    X = rng.uniform(-1, 1, (300, d_in))
    uref = og.forward(flat.astype(np.float64), d_in, widths, torch.float64, X, activation=act)
    assert np.max(np.abs(eng.forward(X.astype(np.float32)).cpu().numpy() - uref[:, 0])) < 3e-6 * max(1, np.max(np.abs(uref)))
    assert np.max(np.abs(eng.forward_f64(X).cpu().numpy() - uref[:, 0])) < 1e-12
    eng.close()
    from varnet_amd.engine import VNEngine, VNError
    with pytest.raises(VNError):
        VNEngine(dim, d_in, widths, True, integNum, kernel=1)     # no generic kernels beyond 6 layers
