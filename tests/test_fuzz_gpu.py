"""Randomised GPU parity: the AUTO kernel choice against the fp64 oracle on random small problems (depth,
ragged widths, d_in, dim, integNum, source / quadrature weights / per-test-function detJ)."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from tests.test_engine_gpu import synth, make_engine, oracle_eval, GRAD_RTOL, LOSS_RTOL  # noqa: E402


@pytest.mark.parametrize('seed', range(40))
def test_random_shape_matches_oracle(seed):
    rng = np.random.default_rng(7000 + seed)
    L = int(rng.integers(1, 7))
    widths = [int(rng.choice([7, 20, 33, 49, 50, 60]))] * L if rng.random() < 0.4 else [int(rng.integers(2, 51 if seed < 24 else 65)) for _ in range(L)]
    dim = int(rng.integers(1, 4))
    d_in = dim + 1 + int(rng.integers(0, 2))
    q = int(rng.choice([4, 8, 16, 27, 36, 64, 216]))
    n_k = int(rng.integers(1, max(2, 2000 // q)))
    nB = int(rng.integers(2, 120)); bDof = int(rng.integers(1, nB))
    src, iw, djv = bool(rng.random() < 0.5), bool(rng.random() < 0.5), bool(rng.random() < 0.3)
    d = synth(7000 + seed, d_in, dim, widths, q, n_k, nB, bDof, src, iw, djv)
    eng = make_engine(d_in, dim, widths, q, src, iw, 0)
    eng.init_params(seed=seed)
    flat = eng.get_params() + 0.05 * rng.standard_normal(eng.P).astype(np.float32)
    eng.set_params(flat)
    eng.set_fe_table(d['N1'], d['dNt1'], d['integW'])
    eng.set_interior(0, d['Input'], d['gcoef'], d['source'], n_k=n_k, detJ=d['detJ'])
    eng.set_bic(d['biInput'], d['biLabel'], bDof, 2.0)
    eng.set_weights(d['w'])
    ref, gref = oracle_eval(flat, d, d_in, dim, widths, q, n_k, bDof, src, iw, djv)
    gb = eng.bind_grad_buffer()
    eng.grad(0)
    torch.cuda.synchronize()
    g = gb.cpu().numpy()
    eng.close()
    info = (L, widths, d_in, dim, q, n_k, nB, src, iw, djv)
    assert abs(g[eng.P] - ref['loss']) <= LOSS_RTOL * abs(ref['loss']), (info, abs(g[eng.P] - ref['loss']) / abs(ref['loss']))
    assert np.max(np.abs(g[:eng.P] - gref)) <= GRAD_RTOL * np.max(np.abs(gref)), info
