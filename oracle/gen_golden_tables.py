"""
ORACLE tooling (test infrastructure): the reference's tabulated exact values (Mojtabi & Deville) -> tests/golden/exact_tables.npz.

The reference holds 25 tabulated values of the 1D+t advection-diffusion solution for two diffusivities:
  * /root/reference/Operator_1Dt.py:113-128      kappa = 0.01/pi (only evaluated when D == 0.01/pi)
  * /root/reference/Operator_1DtMOR.py:117-150   kappa = 0.01/pi (`cExD3`) and kappa = 0.005 (`cExD4`)
and its own acceptance metric l2Err(cEx, cApp) (Operator_1Dt.py:177-186).  Both scripts need IPython and an
interactive matplotlib (Operator_1Dt.py:65-66), so they are not imported: the array LITERALS are read from the
source text with `ast` (data only: numbers), paired exactly as the scripts pair them (uf.pairMats(x, t) with a single
time = one row [x, t] per x; np.vstack of the three snapshots), and the Operator_1Dt copy of the kappa = 0.01/pi table
is checked against the Operator_1DtMOR one.  Runs ONLY in the build container.

Pairing note: Operator_1DtMOR.py:216-224 compares kappa = 0.01/pi against `cExD4` and 0.005
against `cExD3`, i.e. swapped; the fixture stores each table under the diffusivity its comment names
(`# Accuracy points for D = 0.01/pi` -> cExD3, `# Accuracy points for D = 5e-3` -> cExD4).

    python oracle/gen_golden_tables.py
"""
import ast
import os
import sys

sys.dont_write_bytecode = True

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def array_literals(path, names):
    """Every assignment `name = np.array([[...]])[.T]` in the file, in source order -> {name: [arrays]}."""
    tree = ast.parse(open(path).read())
    found = {n: [] for n in names}
    for node in ast.walk(tree):
        if not isinstance(node, ast.Assign) or len(node.targets) != 1 or not isinstance(node.targets[0], ast.Name):
            continue
        name = node.targets[0].id
        if name not in found:
            continue
        v, transpose = node.value, False
        if isinstance(v, ast.Attribute) and v.attr == 'T':
            v, transpose = v.value, True
        if not (isinstance(v, ast.Call) and getattr(v.func, 'attr', '') == 'array'):
            continue
        arr = np.array(ast.literal_eval(v.args[0]), dtype=float)
        found[name].append((node.lineno, arr.T if transpose else arr))
    return {n: [a for _, a in sorted(v, key=lambda p: p[0])] for n, v in found.items()}


def pair(x, t):
    """uf.pairMats(x, t) for a single time row (UtilityFunc.py:301-339): one row [x_i, t] per x_i."""
    assert t.shape == (1, 1)
    return np.hstack([x, np.tile(t, [x.shape[0], 1])])


def tables(path, n_sets):
    lit = array_literals(path, ['xEx1', 'tEx1', 'xEx2', 'tEx2', 'tEx3', 'cEx1', 'cEx2', 'cEx3'])
    x1, x2 = lit['xEx1'][0], lit['xEx2'][0]
    t1, t2, t3 = lit['tEx1'][0], lit['tEx2'][0], lit['tEx3'][0]
    inp = np.vstack([pair(x1, t1), pair(x2, t2), pair(x1, t3)])          # xEx3 = xEx1 in both scripts
    sets = []
    for s in range(n_sets):
        sets.append(np.vstack([lit['cEx1'][s], lit['cEx2'][s], lit['cEx3'][s]]))
    return inp, sets


def main():
    inp_m, (d3, d4) = tables(os.path.join(REF, 'Operator_1DtMOR.py'), 2)
    inp_1, (d3_1,) = tables(os.path.join(REF, 'Operator_1Dt.py'), 1)
    assert inp_m.shape == (25, 2) and d3.shape == (25, 1) and d4.shape == (25, 1)
    assert np.array_equal(inp_m, inp_1) and np.array_equal(d3, d3_1), 'the two scripts hold the same kappa = 0.01/pi table'
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, 'exact_tables.npz'), inpEx=inp_m,
                        cEx_kappa_0p01_over_pi=d3, cEx_kappa_0p005=d4,
                        kappa=np.array([0.01 / np.pi, 0.005]))
    print('wrote exact_tables.npz: 25 points x 2 diffusivities; first row', inp_m[0], d3[0], d4[0])


if __name__ == '__main__':
    main()
