"""
ORACLE tooling (test infrastructure): outputs of the reference's own analytical solution of the 2D+t demo
(/root/reference/Operator_2Dt.py:89-132, `cExFun`: Leij & Dane's solution integrated over time) -> tests/golden/cexfun_2dt.npz.

Operator_2Dt.py itself cannot be imported (IPython magics at :75-76, a hard-coded home folder at :163), so the script is not run:
the parameter assignments (:82-87) and the `cExFun` FunctionDef are cut out of its source with `ast` and executed on their own, with
the names the script binds at its top (:55-66: numpy functions, scipy's erf) and the reference's own, directly importable
`UtilityFunc.UF` (pairMats, nodeNum, isnone).  Runs ONLY in the build container; the .npz (inputs + outputs) is committed.

    MPLBACKEND=Agg python oracle/gen_golden_cexfun.py
"""
import ast
import os
import sys

sys.dont_write_bytecode = True

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def main():
    os.environ.setdefault('MPLBACKEND', 'Agg')
    sys.path.insert(0, REF)
    import UtilityFunc as RU
    from scipy import special
    src = open(os.path.join(REF, 'Operator_2Dt.py')).read()
    tree = ast.parse(src)
    keep = []
    for node in tree.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) \
                and node.targets[0].id in ('T', 'q', 'kappa', 'c0', 'a', 'nt'):
            keep.append(node)
        if isinstance(node, ast.FunctionDef) and node.name == 'cExFun':
            keep.append(node)
    assert [getattr(n, 'name', None) or n.targets[0].id for n in keep] == ['T', 'q', 'kappa', 'c0', 'a', 'nt', 'cExFun']
    ns = {'np': np, 'reshape': np.reshape, 'exp': np.exp, 'pi': np.pi, 'erf': special.erf, 'uf': RU.UF()}
    exec(compile(ast.Module(body=keep, type_ignores=[]), 'Operator_2Dt.py', 'exec'), ns)
    rng = np.random.default_rng(0)
    # points of the rectangle [0,2] x [-.5,.5]: a grid (incl. the inlet x = 0, where the BC is enforced) + random points
    gx, gy = np.meshgrid(np.linspace(0.0, 2.0, 21), np.linspace(-0.5, 0.5, 11))
    x = np.vstack([np.column_stack([gx.ravel(), gy.ravel()]), np.column_stack([rng.uniform(0, 2, 40), rng.uniform(-.5, .5, 40)])])
    # called as the script calls it (Operator_2Dt.py:176: no `t`: every one of the nt = 151 time nodes per point).  The `t`
    # branch (:123-126) hands a 1-D array to uf.nodeNum, which insists on a column matrix: it raises in the reference.
    full = ns['cExFun'](x)                          # [nx, nt]
    np.savez_compressed(os.path.join(OUT, 'cexfun_2dt.npz'), x=x, c_all=full,
                        params=np.array([ns['T'], ns['q'][0], ns['q'][1], ns['kappa'], ns['c0'], ns['a'], ns['nt']]))
    print('wrote cexfun_2dt.npz:', x.shape, full.shape, 'max', full.max(), 'at t=T mean', full[:, -1].mean())


if __name__ == '__main__':
    main()
