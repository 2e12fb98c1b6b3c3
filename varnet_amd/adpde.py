"""
Advection-diffusion PDE container: `ADPDE(domain, diff, vel, source, timeDependent, tInterval,
BCs, IC, cEx, MORvar, d_diff)` -- /root/reference/ADPDE.py:56-246 restated (plot helpers are
out of scope).

    c_t = div(diff grad c) - vel . grad c + source,    a * dc/dn + b * c = g  on each edge.

Constants are wrapped into callables f(x[, t]) returning column arrays; every BC is normalised
to [a, b, g(x,t)] and classified Dirichlet / Neumann / Robin; with a `MOR` instance a lookup
table records which PDE callable each parametric function is.
"""
import numpy as np

from .utility import UF

uf = UF()


class ADPDE:
    def __init__(self, domain, diff, vel, source=0.0, timeDependent=False, tInterval=None,
                 BCs=None, IC=None, cEx=None, MORvar=None, d_diff=None):
        # the reference ignores the `timeDependent` argument (ADPDE.py:108-109)
        timeDependent = tInterval is not None

        if not uf.isnumber(diff) and not callable(diff):
            raise ValueError('diffusivity field must be constant or callable!')
        if not uf.isnumber(vel) and not callable(vel):
            raise ValueError('velocity field must be constant or callable!')
        if not uf.isnumber(source) and not callable(source):
            raise ValueError('source function must be constant or callable!')
        if BCs is not None and not isinstance(BCs, list):
            raise ValueError('BCs must be empty or a list of [a, b, g(x,t)]!')
        if BCs is not None and len(BCs) != domain.bIndNum:
            raise ValueError('number of BCs does not match number of boundaries in domain!')
        if timeDependent and IC is None:
            raise ValueError('initial condition must be provided for time-dependent problems!')
        if cEx is not None and not callable(cEx):
            raise ValueError('exact solution must be a callable function!')
        if d_diff is not None and not uf.isnumber(d_diff) and not callable(d_diff):
            raise ValueError('diffusivity gradient must be constant or callable!')

        dim = domain.dim

        def const_field(val, ncol):
            return lambda x, t=0: val * np.ones([np.shape(x)[0], ncol])

        if callable(diff):
            self.diffFun = diff
        else:
            self.diff = diff
            self.diffFun = const_field(diff, 1)
        if callable(vel):
            self.velFun = vel
        else:
            self.vel = vel
            self.velFun = const_field(np.asarray(vel, dtype=float), dim)
        if callable(source):
            self.sourceFun = source
        else:
            self.source = source
            self.sourceFun = const_field(source, 1)
        if callable(d_diff):
            self.d_diffFun = d_diff
        else:
            d_diff = 0.0 if d_diff is None else d_diff
            self.d_diff = d_diff
            self.d_diffFun = const_field(np.asarray(d_diff, dtype=float), dim)

        # boundary conditions -> [a, b, g]
        bIndNum = domain.bIndNum
        if BCs is None:
            BCs = [[] for _ in range(bIndNum)]
        BCs = list(BCs)
        # Reference quirk kept for parity (ADPDE.py:180-182): the lambda wrapping a constant g
        # closes over a loop variable, so when several BCs carry constant values ALL of them
        # evaluate to the value of the LAST constant BC in the list.
        last_const = {}
        for bInd in range(bIndNum):
            bc = BCs[bInd]
            if uf.isempty(bc):
                BCs[bInd] = [0.0, 1.0, lambda x, t=0: np.zeros([len(x), 1])]
            elif len(bc) != 3:
                raise ValueError('BCs must be specified as a list of [a, b, g(x,t)]!')
            elif not callable(bc[2]):
                last_const['g'] = bc[2]
                BCs[bInd] = [bc[0], bc[1], lambda x, t=0: last_const['g'] * np.ones([len(x), 1])]
        BCtype = []
        for bInd in range(bIndNum):
            if BCs[bInd][0] == 0:
                BCtype.append('Dirichlet')
            elif BCs[bInd][1] == 0:
                BCtype.append('Neumann')
            else:
                BCtype.append('Robin')

        if timeDependent and uf.isempty(IC):
            IC = lambda x, t=0: np.zeros([len(x), 1])
        elif timeDependent and not callable(IC):
            ICval = IC
            IC = lambda x, t=0: ICval * np.ones([len(x), 1])

        # MOR lookup table (ADPDE.py:200-236)
        if MORvar is not None:
            BCind = [None] * bIndNum
            bDataFlg = False
            tab = {'diff': None, 'vel': None, 'source': None, 'IC': None, 'd_diff': None}
            for i, fh in enumerate(MORvar.funcHandles):
                if fh == self.diffFun:
                    tab['diff'] = i
                elif fh == self.velFun:
                    tab['vel'] = i
                elif fh == self.sourceFun:
                    tab['source'] = i
                elif fh == IC:
                    tab['IC'] = i
                elif fh == self.d_diffFun:
                    tab['d_diff'] = i
                else:
                    for bInd in range(bIndNum):
                        if fh == BCs[bInd][2]:
                            BCind[bInd] = i
                            bDataFlg = True
            if tab['diff'] is not None and callable(d_diff) and tab['d_diff'] is None:
                raise ValueError('\'diff\' has extra input arguments but \'d_diff\' does not!')
            tab['BCs'] = BCind
            inp = any(tab[k] is not None for k in ('diff', 'vel', 'source'))
            tab['inpData'] = True if inp else None
            tab['biData'] = True if (bDataFlg or tab['IC'] is not None) else None
            self.MORfunInd = tab

        self.dim = dim
        self.domain = domain
        self.timeDependent = timeDependent
        self.tInterval = tInterval
        self.BCs = BCs
        self.BCtype = BCtype
        self.IC = IC
        self.cEx = cEx
        self.MORvar = MORvar
