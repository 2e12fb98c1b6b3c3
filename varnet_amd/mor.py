"""
Parametric ("model-order-reduction") variable bookkeeping: `MOR(funcHandles, ArgNames,
ArgRange)` with `discretizeArg` / `argIndex` -- /root/reference/MOR.py:52-254 restated
(the unused POD helper, MOR.py:258-320, is out of scope).
"""
import numpy as np

from .utility import UF

uf = UF()


class MOR:
    def __init__(self, funcHandles, ArgNames, ArgRange):
        if not isinstance(funcHandles, list) and callable(funcHandles):
            funcHandles, ArgNames, ArgRange = [funcHandles], [ArgNames], [ArgRange]
        elif not isinstance(funcHandles, list):
            raise ValueError('\'funcHandles\' must be a list of callable functions!')
        ArgNames = list(ArgNames)
        ArgRange = list(ArgRange)
        varNum, argInd, sortInd = [], [], []
        for i, func in enumerate(funcHandles):
            if not callable(func):
                raise ValueError('entries must be callable functions!')
            code = func.__code__
            names = code.co_varnames[:code.co_argcount]
            if not isinstance(ArgNames[i], list):
                ArgNames[i] = [ArgNames[i]]
                ArgRange[i] = [ArgRange[i]]
            pos = []
            for nm in ArgNames[i]:
                if nm not in names:
                    raise ValueError(nm + ' is not an argument of ' + code.co_name + '!')
                pos.append(names.index(nm))
            order = np.argsort(pos)
            pos = uf.reorderList(pos, order)
            ArgNames[i] = uf.reorderList(ArgNames[i], order)
            # variable arguments must be contiguous and last (MOR.py:118-120)
            if not (pos[-1] == code.co_argcount - 1 and len(pos) == pos[-1] - pos[0] + 1):
                raise ValueError('variable arguments of ' + code.co_name +
                                 ' must be ordered and the last arguments to the function')
            if np.shape(ArgRange[i])[1] != 2:
                raise ValueError('dimension of the variable ranges for function ' + code.co_name +
                                 'are not equal to 2!')
            if len(ArgRange[i]) != len(pos):
                raise ValueError('number of variable ranges for function ' + code.co_name +
                                 'does not match the number of variable arguments!')
            ArgRange[i] = uf.reorderList(ArgRange[i], order)
            varNum.append(len(pos))
            argInd.append(pos)
            sortInd.append(order)
        self.funNum = len(funcHandles)
        self.funcHandles = funcHandles
        self.ArgNames = ArgNames
        self.ArgRange = ArgRange
        self.varNum = varNum
        self.argInd = argInd
        self.sortInd = sortInd

    def discretizeArg(self, discScheme, randFlag=False):
        """Per function: matrix of all combinations of its discretised arguments (MOR.py:147-233)."""
        if not isinstance(discScheme, list) and callable(discScheme):
            discScheme = [discScheme]
        elif not isinstance(discScheme, list):
            raise ValueError('\'discScheme\' must be a list!')
        discScheme = list(discScheme)
        for i in range(self.funNum):
            if callable(discScheme[i]):
                continue
            if np.size(discScheme[i]) not in (1, self.varNum[i]):
                raise ValueError('number of discretization numbers for function ' +
                                 self.funcHandles[i].__code__.co_name +
                                 ' does not match the number of variable arguments!')
            if np.size(discScheme[i]) == 1:
                discScheme[i] = np.tile(discScheme[i], self.varNum[i])
        out = []
        for i in range(self.funNum):
            if callable(discScheme[i]):
                vals = discScheme[i]()
                if np.shape(vals)[1] != self.varNum[i]:
                    raise ValueError('output dimension of the function handle to discretize ' +
                                     self.funcHandles[i].__code__.co_name +
                                     ' is not equal to its number of variable arguments!')
                out.append(vals)
                continue
            scheme = uf.reorderList(discScheme[i], self.sortInd[i])
            comb = []
            for j in range(self.varNum[i]):
                lo, hi = self.ArgRange[i][j][0], self.ArgRange[i][j][1]
                n = int(scheme[j])
                disc = np.sort(np.random.uniform(lo, hi, n)) if randFlag else np.linspace(lo, hi, n)
                comb = uf.pairMats(comb, np.reshape(disc, [n, 1]))
            out.append(comb)
        return out

    def argIndex(self, discArg):
        """[nBatches, funNum] index combinations, first function slowest (MOR.py:236-254)."""
        ind = []
        for i in range(self.funNum):
            n = len(discArg[i])
            ind = uf.pairMats(ind, np.reshape(np.arange(n), [n, 1]))
        return ind
