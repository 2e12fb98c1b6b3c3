"""Randomised cross-check of the AUTO kernel choice (fused16 / fused32 / two-pass / generic) against the
generic kernels: random depth, widths (uniform and ragged), d_in, dim, integNum, source / integW / detJvec /
per-row tables, sizes from one tile to several tiles per workgroup.   python tools/fuzz_parity.py [cases] [seed]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from varnet_amd.engine import VNEngine


def synth(seed, d_in, dim, widths, integNum, n_k, nB, bDof, source=False, integW=False, detJvec=False):
    rng = np.random.default_rng(seed)
    n = n_k * integNum
    d = dict(Input=rng.uniform(-1, 1, (n, d_in)).astype(np.float32), gcoef=rng.standard_normal((n, dim)).astype(np.float32),
             source=rng.standard_normal((n, 1)).astype(np.float32) if source else None,
             N1=rng.uniform(0, 1, integNum).astype(np.float32), dNt1=rng.standard_normal(integNum).astype(np.float32),
             integW=rng.uniform(0.5, 1.0, (1, integNum)).astype(np.float32) if integW else None,
             detJ=(rng.uniform(0.1, 0.2, (n_k, 1)).astype(np.float32) if detJvec else np.float32(0.137)),
             biInput=rng.uniform(-1, 1, (nB, d_in)).astype(np.float32), biLabel=rng.standard_normal((nB, 1)).astype(np.float32),
             w=np.array([3.0, 2.0, 5.0]))
    d['N'] = np.tile(d['N1'], n_k).reshape(n, 1)
    d['dNt'] = np.tile(d['dNt1'], n_k).reshape(n, 1)
    return d


def make_engine(d_in, dim, widths, integNum, source, integW, kernel=0, act='sigmoid'):
    return VNEngine(dim, d_in, widths, True, integNum, isSource=source, integWflag=integW, kernel=kernel, activationFun=act)


ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(ncases):
    L = int(rng.integers(1, 7))
    act = 'tanh' if rng.random() < 0.3 else 'sigmoid'
    if rng.random() < 0.5:
        widths = [int(rng.choice([7, 10, 20, 30, 32, 33, 40, 48, 49, 50, 51, 56, 60, 63, 64]))] * L
    else:
        widths = [int(rng.integers(1, 65)) for _ in range(L)]
    dim = int(rng.integers(1, 4)); td = True
    d_in = dim + 1 + int(rng.integers(0, 2))
    q = int(rng.choice([4, 8, 16, 27, 32, 36, 64, 128, 216, 256, 1296]))     # 256: 3D+t 2-point, 1296: 3D+t 3-point Gauss
    big = rng.random() < 0.25
    n_k = int(rng.integers(1, 40)) if not big else int(rng.integers(300, 2500) * 128 // q + 1)
    nB = int(rng.integers(2, 300)); bDof = int(rng.integers(1, nB))
    src, iw, djv = bool(rng.random() < 0.5), bool(rng.random() < 0.5), bool(rng.random() < 0.3)
    d = synth(1000 + case, d_in, dim, widths, q, n_k, nB, bDof, src, iw, djv)
    rows = bool(rng.random() < 0.2)
    grads = []
    try:
        make_engine(d_in, dim, widths, q, src, iw, 1, act).close()
    except Exception as e:                      # deep + wide: too big for the generic kernels' LDS, nothing to cross-check with
        print('case %3d skipped (%s)' % (case, str(e)[:60]), flush=True)
        continue
    for kernel in (1, 0):
        eng = make_engine(d_in, dim, widths, q, src, iw, kernel, act)
        eng.init_params(seed=case)
        flat = eng.get_params() + 0.05 * np.random.default_rng(case).standard_normal(eng.P).astype(np.float32)
        eng.set_params(flat)
        eng.set_fe_table(d['N1'], d['dNt1'], d['integW'])
        kw = dict(N_rows=d['N'], dNt_rows=d['dNt']) if rows else {}
        eng.set_interior(0, d['Input'], d['gcoef'], d['source'], n_k=n_k, detJ=d['detJ'], **kw)
        eng.set_bic(d['biInput'], d['biLabel'], bDof, 2.0)
        eng.set_weights(d['w'])
        gb = eng.bind_grad_buffer(); eng.grad(0); torch.cuda.synchronize()
        grads.append(gb.cpu().numpy().astype(np.float64)); eng.close()
    g0, g1 = grads; P = g0.size - 4
    err = np.max(np.abs(g1[:P] - g0[:P])) / max(np.max(np.abs(g0[:P])), 1e-30)
    lerr = abs(g1[P] - g0[P]) / max(abs(g0[P]), 1e-30)
    worst = max(worst, err, lerr)
    flag = '' if (err < 3e-4 and lerr < 5e-5) else '   <<<<<<<< MISMATCH'
    print('case %3d %s L=%d widths=%s d_in=%d dim=%d q=%d n_k=%d nB=%d src=%d iw=%d djv=%d rows=%d: grad %.1e loss %.1e%s'
          % (case, act, L, widths, d_in, dim, q, n_k, nB, src, iw, djv, rows, err, lerr, flag), flush=True)
    if flag: sys.exit(1)
print('all %d cases agree; worst relative deviation %.2e' % (ncases, worst))
