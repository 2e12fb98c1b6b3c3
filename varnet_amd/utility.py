"""
Hot-path subset of the reference's `UF` helper bag (/root/reference/UtilityFunc.py), restated:
only what the VarNet training path reaches (isnone/isempty/vstack/hstack/pairMats/l2Err/
polyArea/buildDict/reorderList/mergeDict).  Same names and argument meaning so that user
scripts written against the reference keep working.
"""
import numbers
import os

import numpy as np


class UF:
    # -- emptiness / None tests (UtilityFunc.py:101-138) ---------------------------------
    def isnumber(self, x):
        if isinstance(x, (list, tuple, np.ndarray)):
            return all(isinstance(v, numbers.Number) for v in np.asarray(x, dtype=object).reshape(-1))
        return isinstance(x, numbers.Number)

    def isempty(self, x):
        if isinstance(x, (list, dict)):
            return len(x) == 0
        return np.size(x) == 0

    def unpackList(self, x):
        if not isinstance(x, list):
            raise ValueError('Input argument must be a list!')
        out = []
        for item in x:
            if isinstance(item, list):
                out.extend(self.unpackList(item))
            elif not self.isempty(item):
                out.append(item)
        return out

    def isnone(self, x):
        """True if x is None or any element of x is None (empty containers are not None)."""
        if x is None:
            return True
        if self.isempty(x):
            return False
        if isinstance(x, list):
            for item in self.unpackList(x):
                if isinstance(item, np.ndarray):
                    if item.dtype == object and any(v is None for v in item.reshape(-1)):
                        return True
                elif item is None:
                    return True
            return False
        if isinstance(x, np.ndarray):
            return x.dtype == object and any(v is None for v in x.reshape(-1))
        return False

    # -- stacking that tolerates empty operands (UtilityFunc.py:159-184) -------------------
    def vstack(self, tup):
        keep = [t for t in tup if not self.isempty(t)]
        return np.vstack(keep) if keep else []

    def hstack(self, tup):
        keep = [t for t in tup if not self.isempty(t)]
        return np.hstack(keep) if keep else []

    # -- cartesian pairing (UtilityFunc.py:301-339) ----------------------------------------
    def pairMats(self, mat1, mat2, reverse=False):
        """
        Rows of the result are all (row of mat1, row of mat2) pairs, mat1 index slowest.
        With reverse=True the roles are swapped but the column order [mat1 | mat2] is kept.
        """
        if self.isempty(mat1):
            return mat2
        if self.isempty(mat2):
            return mat1
        if reverse:
            mat1, mat2 = mat2, mat1
        mat1 = np.asarray(mat1)
        mat2 = np.asarray(mat2)
        n1, n2 = mat1.shape[0], mat2.shape[0]
        A = np.repeat(mat1, n2, axis=0)
        B = np.tile(mat2, (n1, 1))
        return np.hstack([A, B]) if not reverse else np.hstack([B, A])

    # -- small helpers -------------------------------------------------------------------
    def reorderList(self, x, ind):
        ind = np.reshape(np.asarray(ind), -1)
        return [x[int(i)] for i in ind]

    def buildDict(self, keys, values):
        if len(keys) != len(values):
            raise ValueError('length of the keys and values must match!')
        return {k: v for k, v in zip(keys, values)}

    def mergeDict(self, dictList):
        if not isinstance(dictList, list):
            raise ValueError('input must be a list of dictionaries!')
        out = {}
        for d in dictList:
            out.update(d)
        return out

    def l2Err(self, xTrue, xApp):
        """Normalised l2 error ||xTrue-xApp|| / ||xTrue|| (UtilityFunc.py:485-499)."""
        xTrue = np.reshape(np.asarray(xTrue, dtype=float), -1)
        xApp = np.reshape(np.asarray(xApp, dtype=float), -1)
        if xTrue.size != xApp.size:
            raise ValueError('\'xTrue\' and \'xApp\' must have the same shape!')
        return np.linalg.norm(xTrue - xApp) / np.linalg.norm(xTrue)

    def polyArea(self, x, y=None):
        """Shoelace area (UtilityFunc.py:541-563)."""
        if y is None:
            x = np.asarray(x)
            if x.ndim != 2 or x.shape[1] != 2:
                raise ValueError('input must be 2d!')
            x, y = x[:, 1], x[:, 0]
        else:
            x = np.reshape(x, -1)
            y = np.reshape(y, -1)
            if len(x) != len(y):
                raise ValueError('\'x\' and \'y\' must be the same length!')
        return 0.5 * np.abs(np.dot(x, np.roll(y, 1)) - np.dot(y, np.roll(x, 1)))

    # -- segmentation / rejection sampling (UtilityFunc.py:342-451): host-side sampling policy of
    #    smpScheme='optimal' --------------------------------------------------------------
    def listSegment(self, vec, segdof, func=None):
        """Cut `vec` into consecutive segments of `segdof` rows (a trailing remainder becomes one
        more segment); optionally map `func(segment, index)` over them."""
        n = len(vec)
        if segdof is None:
            return [vec] if func is None else [func(vec, 0)]
        if isinstance(segdof, numbers.Number):
            segdof = [segdof]
        segdof = [int(v) for v in segdof]
        if segdof[-1] > n:
            raise ValueError('\'segdof\' is out of bound!')
        out, ind, i = [], 0, 0
        for i, cnt in enumerate(segdof):
            seg = vec[ind:ind + cnt]
            out.append(seg if func is None else func(seg, i))
            ind += cnt
        if ind < n:
            seg = vec[ind:]
            out.append(seg if func is None else func(seg, i))
        return out

    def rejectionSampling(self, func, smpfun, dof, dofT=None):
        """
        Draw `dof[i]` samples per segment with acceptance probability func/max(func over the
        reference grid): `func()` gives the values on the fixed grid, `func(samples)` on
        candidates produced by `smpfun()` (UtilityFunc.py:342-404).
        """
        if isinstance(dof, numbers.Number):
            if dofT is not None:
                raise ValueError('\'dofT\' must be None for scalar \'dof\'')
            dof = [dof]
        dof = [int(v) for v in dof]
        m = len(dof)
        if m > 1 and dofT is None:
            raise ValueError('\'dofT\' must be provided when \'dof\' is a list!')
        fmax = self.listSegment(func(), dofT, lambda x, i: np.max(x))
        kept = [[] for _ in range(m)]
        ns = [0] * m
        again = True
        while again:
            samples = smpfun()
            smpList = self.listSegment(samples, dofT)
            val = func(samples)

            def accept(v, i):
                u = np.random.uniform(size=[len(v), 1])
                return np.reshape(u < (v / fmax[i]), len(v))

            ind = self.listSegment(val, dofT, accept)
            again = False
            for i in range(m):
                kept[i] = self.vstack([kept[i], smpList[i][ind[i]]])
                ns[i] += int(np.sum(ind[i]))
                if ns[i] < dof[i]:
                    again = True
        return np.vstack([kept[i][:dof[i], :] for i in range(m)])

    def mat2diffpack(self, filename, fieldname, mat):
        """Write a vector or matrix as the MATLAB-style m-file Diffpack reads (UtilityFunc.py:829-886):
        `datatype`, the sizes, a `zeros(...)` allocation and one assignment per entry, column by column."""
        if not isinstance(filename, str):
            raise TypeError('\'filename\' must be a string!')
        if '.m' not in filename:
            raise ValueError('\'filename\' must end with \'.m\'!')
        if not isinstance(fieldname, str):
            raise TypeError('\'fieldname\' must be a string!')
        if not isinstance(mat, (list, np.ndarray)):
            raise TypeError('\'mat\' must be an array!')
        mat = np.asarray(mat)
        sh = mat.shape
        if len(sh) > 2:
            raise ValueError('\'mat\' must be at most 2d!')
        out = ['datatype = \'real\';\n\n']
        if len(sh) == 1 or sh[1] == 1:
            vec = mat.reshape(-1)
            out.append('length = %d;\n\n%s = zeros(length,1);\n\n%%%% Data:\n\n' % (sh[0], fieldname))
            out.extend('%s(%d) = \t%s;\n' % (fieldname, i + 1, str(vec[i])) for i in range(sh[0]))
        else:
            out.append('nrows = %d; ncolumns = %d;\nnentries = %d;\n\n' % (sh[0], sh[1], sh[0] * sh[1]))
            out.append('%s = zeros(nrows, ncolumns);\n\n%%%% Data:\n\n' % fieldname)
            for j in range(sh[1]):
                out.extend('%s(%d,%d) = \t%s;\n' % (fieldname, i + 1, j + 1, str(mat[i, j])) for i in range(sh[0]))
        with open(filename, 'w') as f:
            f.write(''.join(out))

    def clearFolder(self, folderpath, ask=True):
        """Empty `folderpath` (files and sub-folders); like the reference (UtilityFunc.py:502-523) it asks on the console
        first unless `ask=False`.  The operator scripts call it before `train` (Operator_1Dt.py:167)."""
        import shutil
        entries = os.listdir(folderpath)
        if not entries:
            return
        while ask:
            answer = input('clear the content of the folder? (y/n)\n').lower()
            if answer in ('y', 'yes'):
                break
            if answer in ('n', 'no'):
                return
        for name in entries:
            path = os.path.join(folderpath, name)
            try:
                if os.path.isdir(path) and not os.path.islink(path):
                    shutil.rmtree(path)
                else:
                    os.remove(path)
            except OSError as e:
                print(e)

    def copyFile(self, filename, folderpath):
        """Back up a file (the operator script itself, Operator_1Dt.py:168) into `folderpath` (UtilityFunc.py:526-538)."""
        import shutil
        if not os.path.exists(filename):
            filename = os.path.join(os.getcwd(), filename)
            if not os.path.exists(filename):
                raise ValueError('The file does not exist!')
        shutil.copy2(filename, folderpath)

    def nodeNum(self, x, val):
        """Index of the entry of x closest to each value in val."""
        x = np.reshape(np.asarray(x, dtype=float), -1)
        val = np.reshape(np.asarray(val, dtype=float), -1)
        return np.array([int(np.argmin(np.abs(x - v))) for v in val])
