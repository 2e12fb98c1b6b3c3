"""
Reference-element tables for the compactly supported multilinear (hat) test functions and their
Gauss-Legendre quadrature -- the `FE` class of /root/reference/FiniteElement.py:52-434,
re-derived as closed-form tensor products (no recursion), with identical orderings:

  * basis / corner multi-index b = (s_0,...,s_{D-1}), s in {-1,+1}, first coordinate slowest
    (FiniteElement.py:126-153);
  * quadrature point q likewise, first coordinate slowest (FiniteElement.py:156-190);
  * a test function centred at a training point is integrated over its 2^D surrounding
    elements; element e and basis b share the numbering (FiniteElement.py:298-324), so the
    per-test-function tables have integNum = 2^D * ip^D entries ordered (element, point).

What the device needs from this class is only the period-`integNum` tables
`N`, `dN`, `delta`, `integW` (see `basisTable`): the reference tiles them to all nT rows on the
host (FiniteElement.py:426-432); here they stay 16..216-entry tables that the kernels index by
`row % integNum`.
"""
import numpy as np


class FE:
    def __init__(self, dim=2, integPnum=2):
        if integPnum > 3:
            raise ValueError('higher order integration needs code modification!')
        if dim not in (1, 2, 3):
            raise ValueError('FE dimension must be 1, 2 or 3!')
        if integPnum == 2:
            integP = np.array([-1.0, 1.0]) / np.sqrt(3.0)
            integW = np.ones(2)
        elif integPnum == 3:
            integP = np.sqrt(3.0 / 5.0) * np.array([-1.0, 0.0, 1.0])
            integW = np.array([5.0, 8.0, 5.0]) / 9.0
        else:
            raise ValueError('integPnum must be 2 or 3!')
        self.dim = dim
        self.basisNum = 2 ** dim
        self.nodeNum = 3 ** dim
        self.integPnum = integPnum
        self.integP = integP
        self.integW = integW
        self.IntegPnum = integPnum ** dim
        self.basMultiInd = self.basisOrder()
        self.IntegP = self.integPoint()
        self.basVal = self.basisVal()
        self.basDeriVal = self.basisDeriVal()
        self.elemCoord = self.elemTranslation()
        self.delta = self.integPtranslation()
        self.IntegW = self.integWeight()
        self.massVec, self.massDelta = self.massVector()

    # ------------------------------------------------------------------------------------
    @staticmethod
    def _grid(vals, dim):
        """All dim-tuples over `vals`, first coordinate slowest.  [len(vals)**dim, dim]."""
        g = np.meshgrid(*([np.asarray(vals)] * dim), indexing='ij')
        return np.stack([a.reshape(-1) for a in g], axis=1)

    def basisOrder(self):
        return self._grid([-1, 1], self.dim).astype(int)

    def integPoint(self):
        return self._grid(self.integP, self.dim).astype(float)

    def _factors(self):
        """f[b, q, d] = 0.5*(1 + s_{b,d} * xi_{q,d})  (1-D hat factor)."""
        s = self.basMultiInd[:, None, :].astype(float)
        xi = self.IntegP[None, :, :]
        return 0.5 * (1.0 + s * xi)

    def basisVal(self):
        """[2^D, ip^D]: basis b at quadrature point q (FiniteElement.py:223-240)."""
        return np.prod(self._factors(), axis=2)

    def basisDeriVal(self):
        """[D, 2^D, ip^D]: d basis_b / d xi_d at q (FiniteElement.py:274-295)."""
        f = self._factors()
        D = self.dim
        out = np.empty((D, self.basisNum, self.IntegPnum))
        for d in range(D):
            others = np.prod(np.delete(f, d, axis=2), axis=2) if D > 1 else np.ones(f.shape[:2])
            out[d] = 0.5 * self.basMultiInd[:, d][:, None] * others
        return out

    def elemTranslation(self):
        """
        [D, 2^D, 2^D]: elemCoord[d, e, i] = coordinate d of corner i of element e, in units of
        the element size, with the training point at the origin: 0.5*(s_{i,d} - s_{e,d})
        (FiniteElement.py:298-324).
        """
        s = self.basMultiInd.astype(float)
        return 0.5 * (s.T[:, None, :] - s.T[:, :, None])

    def integPtranslation(self):
        """[D, 2^D, ip^D]: offsets of the quadrature points from the training point, in units
        of h (isoparametric map, FiniteElement.py:327-348)."""
        return np.einsum('dei,iq->deq', self.elemCoord, self.basVal)

    def integWeight(self):
        """[2^D, ip^D] quadrature weights, or None when all are 1 (FiniteElement.py:351-389)."""
        if np.all(self.integW == 1.0):
            return None
        w = np.prod(self._grid(self.integW, self.dim), axis=1)
        return np.repeat(w[None, :], self.basisNum, axis=0)

    def massVector(self):
        """
        Row of the mass matrix that belongs to the training point (FiniteElement.py:438-499; kept as the attributes
        `massVec [3^D]`, `massDelta [D, 3^D]`, the reference's alternative treatment of nodal source values; nothing
        on the training path uses it): R_j = sum over the 2^D elements around the point of int N N_j, without |J|.
        Element e and its corner i meet at the node with offsets elemCoord[:, e, i] in {-1, 0, 1}^D; node index =
        sum_d (offset_d + 1) * 3^d  (the reference's Fortran-order index matrix).
        """
        w = np.ones(self.IntegPnum) if self.IntegW is None else self.IntegW[0, :]
        R = np.einsum('q,eq,iq->ei', w, self.basVal, self.basVal)             # single-element mass matrix
        idx = self.elemCoord.astype(int) + 1                                   # [D, e, i] in {0, 1, 2}
        node = np.tensordot(3 ** np.arange(self.dim), idx, axes=(0, 0))        # [e, i]
        mvec = np.zeros(self.nodeNum)
        np.add.at(mvec, node.reshape(-1), R.reshape(-1))
        mdelta = np.zeros((self.dim, self.nodeNum))
        mdelta[:, node.reshape(-1)] = self.elemCoord.reshape(self.dim, -1)
        return mvec, mdelta

    # ------------------------------------------------------------------------------------
    def basisTable(self, hVec):
        """
        Period-integNum tables for elements of size hVec (one per FE dimension):
        integNum, detJ, delta [D, integNum], intWeight [1, integNum]|None, N [integNum],
        dN [integNum, D]  (one period of what FiniteElement.py:392-434 tiles).
        """
        hVec = np.reshape(np.asarray(hVec, dtype=float), (self.dim, 1))
        integNum = self.basisNum * self.IntegPnum
        detJ = np.prod(0.5 * hVec)
        delta = self.delta.reshape(self.dim, integNum)
        intWeight = None if self.IntegW is None else self.IntegW.reshape(1, integNum)
        N = self.basVal.reshape(integNum)
        dN = (2.0 / hVec * self.basDeriVal.reshape(self.dim, integNum)).T
        return integNum, detJ, delta, intWeight, N, dN

    def basisTot(self, nt, hVec):
        """Reference signature (FiniteElement.py:392-434): tables tiled to all nt test functions."""
        integNum, detJ, delta, intWeight, N, dN = self.basisTable(hVec)
        nT = nt * integNum
        return (integNum, nT, detJ, delta, intWeight,
                np.tile(N.reshape(integNum, 1), (nt, 1)), np.tile(dN, (nt, 1)))
