"""
ORACLE (test infrastructure, not product code).

Hand-derived NumPy statement of the arithmetic the HIP kernels perform: a forward pass that
carries (value, one directional tangent) through the sigmoid MLP, the weak-form epilogue, and
the matching reverse pass to parameter gradients.  It exists to (a) prove, against the
autograd restatement of the reference graph in oracle/tf1_graph.py, that the
single-tangent formulation is mathematically the reference's
`tf.gradients(model(Input), Input)` contracted with `gcoef` (TFModel.py:536,653-654) followed
by `compute_gradients` (TFModel.py:709); (b) document the kernel math in executable form.

Key identity: the loss only uses  sum_d dM_dx[:,d]*gcoef[:,d]  (TFModel.py:653-654), i.e. the
directional derivative of the network along  gdir = (gcoef_0..gcoef_{dim-1}, 0, ..., 0).
So one forward tangent suffices whatever `dim` is.
"""
import numpy as np


def sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


def split_params(flat, d_in, widths):
    dims = []
    fi = d_in
    for h in widths:
        dims.append((fi, h))
        fi = h
    dims.append((fi, 1))
    out, off = [], 0
    for i, o in dims:
        W = flat[off:off + i * o].reshape(i, o)
        off += i * o
        b = flat[off:off + o]
        off += o
        out.append((W, b))
    return out


def forward_tangent(params, X, G):
    """X [n,d_in] points, G [n,d_in] tangent direction.  Returns u, udot, cache."""
    a, ad = X, G
    cache = []
    for W, b in params[:-1]:
        z = a @ W + b
        zd = ad @ W
        an = sigmoid(z)
        sp = an * (1.0 - an)
        cache.append((a, ad, an, zd))
        a, ad = an, sp * zd
    W, b = params[-1]
    u = a @ W + b
    ud = ad @ W
    cache.append((a, ad))
    return u[:, 0], ud[:, 0], cache


def backward(params, cache, ubar, udbar):
    """
    Reverse of forward_tangent.  ubar/udbar [n]: d loss / d u, d loss / d udot.
    Returns list of (gW, gb) in layer order.
    """
    grads = [None] * len(params)
    a, ad = cache[-1]
    W, b = params[-1]
    grads[-1] = (a.T @ ubar[:, None] + ad.T @ udbar[:, None], np.array([ubar.sum()]))
    abar = ubar[:, None] * W[:, 0][None, :]
    adbar = udbar[:, None] * W[:, 0][None, :]
    for l in range(len(params) - 2, -1, -1):
        W, b = params[l]
        ap, adp, an, zd = cache[l]
        sp = an * (1.0 - an)
        spp = sp * (1.0 - 2.0 * an)
        zdbar = adbar * sp
        zbar = abar * sp + adbar * spp * zd
        grads[l] = (ap.T @ zbar + adp.T @ zdbar, zbar.sum(axis=0))
        abar = zbar @ W.T
        adbar = zdbar @ W.T
    return grads


def loss_and_grad(flat, d_in, widths, dim, Input, gcoef, source, N, dNt, integW, n_k,
                  integNum, detJ, biInput, biLabel, bDof, biDimVal, w, time_dependent=True):
    """
    Same contract as oracle.tf1_graph.loss_and_grad (detJ scalar or [n_k]); everything in the
    dtype of `flat`.  Returns (dict, flat gradient).
    """
    dt = flat.dtype
    params = split_params(flat, d_in, widths)
    n = Input.shape[0]
    G = np.zeros((n, d_in), dtype=dt)
    G[:, :dim] = gcoef
    u, ud, cache = forward_tangent(params, Input.astype(dt), G)
    int1 = ud.copy()
    if time_dependent:
        int1 -= u * dNt.reshape(-1)
    if source is not None:
        int1 -= source.reshape(-1) * N.reshape(-1)
    wq = np.ones(integNum, dtype=dt) if integW is None else integW.reshape(-1).astype(dt)
    int1 = int1.reshape(n_k, integNum) * wq[None, :]
    R = int1.sum(axis=1)
    detJk = np.broadcast_to(np.asarray(detJ, dtype=dt).reshape(-1), (n_k,)) if np.size(detJ) > 1 \
        else np.full(n_k, detJ, dtype=dt)
    lossVec = detJk * R * R
    varLoss = lossVec.sum()
    seed = (2.0 * w[2] * detJk * R)[:, None] * wq[None, :]
    seed = seed.reshape(-1)
    ubar = -dNt.reshape(-1) * seed if time_dependent else np.zeros_like(seed)
    grads = backward(params, cache, ubar, seed)

    nB = 0 if biInput is None else biInput.shape[0]
    bc = ic = dt.type(0.0)
    if nB:
        ub, _, cb = forward_tangent(params, biInput.astype(dt), np.zeros((nB, d_in), dtype=dt))
        e = ub - biLabel.reshape(-1)
        bc = biDimVal * np.mean(e[:bDof] ** 2)
        coef = np.empty(nB, dtype=dt)
        coef[:bDof] = 2.0 * w[0] * biDimVal / bDof
        if time_dependent:
            ic = biDimVal * np.mean(e[bDof:] ** 2)
            coef[bDof:] = 2.0 * w[1] * biDimVal / (nB - bDof)
        else:
            coef[bDof:] = 0.0
        gb = backward(params, cb, coef * e, np.zeros(nB, dtype=dt))
        grads = [(a[0] + b_[0], a[1] + b_[1]) for a, b_ in zip(grads, gb)]
    loss = w[0] * bc + w[1] * ic + w[2] * varLoss
    g = np.concatenate([np.concatenate([gW.reshape(-1), gb_.reshape(-1)]) for gW, gb_ in grads])
    return dict(loss=loss, BCloss=bc, ICloss=ic, varLoss=varLoss, lossVec=lossVec), g
