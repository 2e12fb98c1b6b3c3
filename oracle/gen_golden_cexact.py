"""
ORACLE tooling (test infrastructure): outputs of the reference's own Fourier-series solution of the 1D+t demo
(/root/reference/Operator_1Dt.py:78-108 `cExact`, the known answer of BASELINE config 1's problem, D = 0.1/pi; and the
diffusivity-parametrised copy /root/reference/Operator_1DtMOR.py:77-110 that the MOR script evaluates at D = 0.1/pi and
D = 0.1, :226-234) -> tests/golden/cexact_1dt.npz.

Neither script can be imported (IPython magics, hard-coded home folders), so they are not run: the parameter assignments
and the `IC` / `cExact` FunctionDefs are cut out of their source with `ast` and executed on their own with the names the
scripts bind at their top (numpy functions).  Runs ONLY in the build container; the .npz (inputs + outputs) is committed.

    python oracle/gen_golden_cexact.py
"""
import ast
import os
import sys

sys.dont_write_bytecode = True

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def cut(script, assigns, funcs):
    tree = ast.parse(open(os.path.join(REF, script)).read())
    keep = []
    for node in tree.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) \
                and node.targets[0].id in assigns:
            keep.append(node)
        if isinstance(node, ast.FunctionDef) and node.name in funcs:
            keep.append(node)
    got = [getattr(n, 'name', None) or n.targets[0].id for n in keep]
    assert got == list(assigns) + list(funcs), got
    ns = {'np': np, 'reshape': np.reshape, 'exp': np.exp, 'pi': np.pi, 'sin': np.sin, 'cos': np.cos, 'shape': np.shape}
    exec(compile(ast.Module(body=keep, type_ignores=[]), script, 'exec'), ns)
    return ns


def main():
    one = cut('Operator_1Dt.py', ('u', 'D', 'T'), ('IC', 'cExact'))
    mor = cut('Operator_1DtMOR.py', ('u', 'T'), ('IC', 'cExact'))
    assert one['D'] == 0.1 / np.pi and one['u'] == 1.0 and one['T'] == 2.0
    rng = np.random.default_rng(0)
    # (1) scattered points of [-1,1] x [0,T] incl. the initial line t = 0 (where the function returns IC) and both boundaries
    gx, gt = np.meshgrid(np.linspace(-1.0, 1.0, 21), np.linspace(0.0, 2.0, 11))
    X = np.vstack([np.column_stack([gx.ravel(), gt.ravel()]), np.column_stack([rng.uniform(-1, 1, 60), rng.uniform(0, 2, 60)])])
    x, t = X[:, 0:1].copy(), X[:, 1:2].copy()
    c_pts = one['cExact'](x, t)
    c_pts_mor = mor['cExact'](x.copy(), t.copy(), D=0.1 / np.pi)
    c_pts_mor_01 = mor['cExact'](x.copy(), t.copy(), D=0.1)
    # (2) the grid the script's acceptance metric uses for D = 0.1/pi (Operator_1Dt.py:177-180): fixData.cEx = cExact on
    # fixData.uniform_input = pairMats(mesh.coordinates, t_coord) of discNum = 20, tDiscNum = 300 (VarNetUtility.py:318, 390).
    # The coordinates are rebuilt here from the formulas (Domain.py:663-668, VarNet.py:326-333), not from our package.
    n, tdof = 20, 300
    h = 2.0 / (n + 1)
    xc = np.linspace(-1.0 + h, 1.0 - h, n).reshape(n, 1)
    ht = 2.0 / tdof
    tc = np.linspace(0.0 + ht, 2.0, tdof).reshape(tdof, 1)
    sys.path.insert(0, REF)
    os.environ.setdefault('MPLBACKEND', 'Agg')
    import UtilityFunc as RU
    ruf = RU.UF()
    ui = ruf.pairMats(xc, tc)
    c_uniform = one['cExact'](ui[:, 0:1], ui[:, 1:2])
    # (3) the grid of the MOR script's metric (Operator_1DtMOR.py:208-210, 226-234): 100 interior nodes x linspace(0,T,100)
    h2 = 2.0 / 101
    xm = np.linspace(-1.0 + h2, 1.0 - h2, 100).reshape(100, 1)
    tm = np.linspace(0, 2.0, num=100).reshape(100, 1)
    inp_m = ruf.pairMats(xm, tm)
    c_m = mor['cExact'](x=inp_m[:, :1], t=inp_m[:, 1:2], D=0.1 / np.pi)
    c_m01 = mor['cExact'](x=inp_m[:, :1], t=inp_m[:, 1:2], D=0.1)
    # the function refuses diffusivities below 0.1/pi (Operator_1DtMOR.py:86-87): recorded as a fact of the reference
    try:
        mor['cExact'](x.copy(), t.copy(), D=0.01)
        refuses = False
    except ValueError:
        refuses = True
    np.savez_compressed(os.path.join(OUT, 'cexact_1dt.npz'), x=x, t=t, c=c_pts, c_mor_D_0p1_over_pi=c_pts_mor, c_mor_D_0p1=c_pts_mor_01,
                        uniform_input=ui, c_uniform=c_uniform, mor_input=inp_m, c_mor_grid_D_0p1_over_pi=c_m, c_mor_grid_D_0p1=c_m01,
                        params=np.array([one['u'], one['D'], one['T']]), mor_refuses_small_D=np.array(refuses))
    print('wrote cexact_1dt.npz:', c_pts.shape, c_uniform.shape, c_m.shape, 'range', c_uniform.min(), c_uniform.max(),
          'max |1Dt - MOR copy|', np.abs(c_pts - c_pts_mor).max(), 'refuses', refuses)


if __name__ == '__main__':
    main()
