"""Soak of the point kernels (not collected by pytest; it uses the oracle, so it lives under tests/):
    python -m tests.soak_points [seeds] [first_seed]
Runs the body of tests/test_fuzz_points_gpu.py over many seeds -- random networks of the 8-wave family, random points -- and reports,
per kernel family that served the network (the bf16-piece kernels of vn_split16.hip at hidden widths 33..64 with 2..7 layers, else the
f32-MFMA kernels vn_pgrad16 / vn_taylor16), the worst errors against the fp64 oracle at the test's bars (u 2e-6, grad u 2e-5,
residual 1e-4 of their own scales).  Round 6: profiles/r6_points_soak.txt."""
import sys

import numpy as np
import torch

from oracle import tf1_graph as og
from tests.test_fuzz_points_gpu import _draw


def served_by_split(widths):
    return 33 <= max(widths) <= 64 and 2 <= len(widths) <= (7 if max(widths) <= 50 else 6)


def main():
    from varnet_amd.engine import VNEngine
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    worst = {True: {'u': 0.0, 'grad': 0.0, 'res': 0.0, 'n': 0}, False: {'u': 0.0, 'grad': 0.0, 'res': 0.0, 'n': 0}}
    f32_same = {'u': 0.0, 'grad': 0.0, 'res': 0.0, 'n': 0}     # round 5's f32-MFMA kernels on the networks the bf16-piece kernels serve
    worse = {'u': 0, 'grad': 0, 'res': 0}                     # cases where the bf16-piece kernel's error is the larger of the two
    bad = 0
    for seed in range(first, first + nseeds):
        rng = np.random.default_rng(seed)
        for case in range(12):
            L, widths, dim, d_in, act, n = _draw(rng)
            eng = VNEngine(dim, d_in, widths, True, 16, activationFun=act)
            if not eng.dedup_supported():
                eng.close()
                continue
            eng.init_params(seed=case)
            flat = (eng.get_params() * float(rng.uniform(1.0, 2.5))).astype(np.float32)
            eng.set_params(flat)
            X = rng.uniform(-1.2, 1.2, (n, d_in))
            diff = rng.uniform(0.05, 1, (n, 1)); vel = rng.standard_normal((n, dim))
            src = rng.standard_normal((n, 1)); ddx = rng.standard_normal((n, dim))
            f64 = flat.astype(np.float64)
            uref, rref = og.residual(f64, d_in, widths, torch.float64, X, diff, vel, src, ddx, dim, True, activation=act)
            params = og.unflatten(f64, d_in, widths, torch.float64)
            Xt = torch.tensor(X, requires_grad=True)
            _, gref, _, _ = og.model_grad(params, Xt, dim, activation=act)
            gref = gref.detach().numpy()
            X32 = X.astype(np.float32)
            u, g = eng.forward_grad(X32)
            uf = eng.forward(X32)
            _, r = eng.residual(X32, diff, vel, src, ddx, fp64=False)
            torch.cuda.synchronize()
            su, sg, sr = max(1.0, np.abs(uref).max()), max(1e-30, np.abs(gref).max()), max(1.0, np.abs(rref).max())
            e = {'u': max(np.abs(u.cpu().numpy() - uref[:, 0]).max(), np.abs(uf.cpu().numpy() - uref[:, 0]).max()) / su,
                 'grad': np.abs(g.cpu().numpy() - gref).max() / sg, 'res': np.abs(r.cpu().numpy() - rref[:, 0]).max() / sr}
            sp = served_by_split(widths)
            ok = e['u'] <= 2e-6 and e['grad'] <= 2e-5 and e['res'] <= 1e-4 and all(np.isfinite(v) for v in e.values())
            ef = None
            if sp:
                # the f32-MFMA kernels (round 5's) on the same inputs: an engine of the cross-check library, route 2
                ex = VNEngine(dim, d_in, widths, True, 16, activationFun=act, xcheck=True)
                ex.set_params(flat)
                ex.debug_point_route(2)
                u2, g2 = ex.forward_grad(X32)
                uf2 = ex.forward(X32)
                _, r2 = ex.residual(X32, diff, vel, src, ddx, fp64=False)
                torch.cuda.synchronize()
                ef = {'u': max(np.abs(u2.cpu().numpy() - uref[:, 0]).max(), np.abs(uf2.cpu().numpy() - uref[:, 0]).max()) / su,
                      'grad': np.abs(g2.cpu().numpy() - gref).max() / sg, 'res': np.abs(r2.cpu().numpy() - rref[:, 0]).max() / sr}
                ex.close()
                for k in ('u', 'grad', 'res'):
                    f32_same[k] = max(f32_same[k], float(ef[k]))
                    if e[k] > ef[k]:
                        worse[k] += 1
                f32_same['n'] += 1
            if not ok:
                bad += 1
                print('OVER THE BAR: seed %d case %d %s widths=%s d_in=%d dim=%d n=%d split=%s: %s%s'
                      % (seed, case, act, widths, d_in, dim, n, sp, {k: '%.2e' % v for k, v in e.items()},
                         '' if ef is None else '   f32-MFMA kernels on the same inputs: %s' % {k: '%.2e' % v for k, v in ef.items()}), flush=True)
            w = worst[sp]
            w['n'] += 1
            for k in ('u', 'grad', 'res'):
                w[k] = max(w[k], float(e[k]))
            eng.close()
        if (seed - first) % 10 == 9:
            print('after seed %d: bf16-piece kernels %s | f32-MFMA kernels %s' % (seed, worst[True], worst[False]), flush=True)
    print('networks served by the bf16-piece kernels: %(n)d, worst u %(u).2e grad %(grad).2e residual %(res).2e' % worst[True])
    print('   the f32-MFMA kernels on those same %(n)d networks:  worst u %(u).2e grad %(grad).2e residual %(res).2e' % f32_same)
    print('   cases where the bf16-piece kernel has the larger error of the two: u %(u)d, grad %(grad)d, residual %(res)d' % worse)
    print('networks served by the f32-MFMA kernels:   %(n)d, worst u %(u).2e grad %(grad).2e residual %(res).2e' % worst[False])
    print('cases over the bars (u 2e-6, grad 2e-5, residual 1e-4): %d' % bad)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
