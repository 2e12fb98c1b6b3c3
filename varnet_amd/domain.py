"""
Spatial domains and their structured discretisation: `Domain1D`, `PolygonDomain2D`, `Mesh`
(/root/reference/Domain.py:55-742), restated with the same constructor signatures and the same
array layouts:

  * interior nodes  linspace(lo+h, hi-h, n), h = (hi-lo)/(n+1)      (Domain.py:459-467, 670-677)
  * 2-D grid is x-fastest, filtered by point-in-polygon              (Domain.py:474-481)
  * boundary nodes  ceil(bDiscNum*edge length) per edge, both ends   (Domain.py:499-519)
"""
import math
import numpy as np

from .utility import UF

uf = UF()


class Domain:
    def __init__(self, dim, lim):
        self.dim = dim
        self.lim = np.array(lim)

    def scaleCoord(self, x):
        """Centre and scale coordinates to [-1,1] (Domain.py:72-90)."""
        x = np.asarray(x)
        if x.shape[1] != self.dim:
            raise ValueError('Input dimensions are incompatible with domain dimension!')
        cen = np.mean(self.lim, axis=0)
        scale = np.diff(self.lim, axis=0)
        return (x - cen) / scale * 2

    def isInside(self, x):
        raise Exception('This function must be redefined in the subclass!')

    def getMesh(self):
        raise Exception('This function must be redefined in the subclass!')


class Mesh:
    """Record of a discretised domain (Domain.py:119-159)."""

    def __init__(self, dim, dof, coordinates, he, bIndNum, bdof, bCoordinates, discNum=[],
                 bDiscNum=[]):
        self.dim = dim
        self.dof = dof
        self.coordinates = coordinates
        self.he = he
        self.bIndNum = bIndNum
        self.bdof = bdof
        self.bCoordinates = bCoordinates
        self.discNum = discNum
        self.bDiscNum = bDiscNum


def _axis_nodes(lo, hi, dof, rfrac, sortflg, tol):
    """Uniform (and optionally random) interior nodes of one axis."""
    dof1 = math.floor(dof * rfrac)
    dof2 = dof - dof1
    c1 = np.random.uniform(lo + tol, hi - tol, dof1)
    c2 = np.linspace(lo + tol, hi - tol, dof2)
    c = np.hstack([c1, c2]) if dof1 else c2
    if rfrac > 0 and sortflg:
        c = np.sort(c)
    return c


class PolygonDomain2D(Domain):
    """Polygon with optional polygonal obstacles; vertices one per row (Domain.py:216-582)."""

    def __init__(self, vertices=np.array([[-1.0, -1.0], [1.0, -1.0], [1.0, 1.0], [-1.0, 1.0]]),
                 obsVertices=[]):
        dim = 2
        vertices = np.asarray(vertices, dtype=float)
        if vertices.shape[1] != dim:
            raise ValueError('Vertex dimensions are incompatible with domain dimension!')
        if not isinstance(obsVertices, list):
            raise ValueError('obstacle polygons must be given as a list of matrices!')
        lim = np.vstack([vertices.min(axis=0), vertices.max(axis=0)])
        super().__init__(dim, lim)
        bGeom = self.boundaryLims(vertices)
        bIndNum = vertices.shape[0]
        for obs in obsVertices:
            bIndNum += np.shape(obs)[0]
            bGeom = np.vstack([bGeom, self.boundaryLims(np.asarray(obs, dtype=float))])
        self.vertexNum = vertices.shape[0]
        self.vertices = vertices
        self.obsNum = len(obsVertices)
        self.obsVertices = obsVertices
        self.bIndNum = bIndNum
        self.boundryGeom = bGeom
        self.measure = uf.polyArea(vertices)

    def boundaryLims(self, vertices):
        """[nEdges,2,2]: end points of every edge, in polygon order (Domain.py:275-286)."""
        v = np.asarray(vertices, dtype=float)
        return np.stack([v, np.roll(v, -1, axis=0)], axis=1)

    def isInside(self, x, tol=0.):
        """Point-in-polygon with matplotlib.path, as the reference does (Domain.py:354-384)."""
        from matplotlib.path import Path
        x = np.asarray(x)
        if x.shape[1] != self.dim:
            raise ValueError('Vertex dimensions are incompatible with domain dimension!')
        inDom = Path(self.vertices, closed=False).contains_points(x, radius=tol)
        if self.obsNum == 0:
            return inDom
        inObs = np.zeros((x.shape[0], self.obsNum), dtype=bool)
        for o in range(self.obsNum):
            inObs[:, o] = Path(self.obsVertices[o], closed=False).contains_points(x, radius=-tol)
        # NB: the reference keeps a point unless it lies in ALL obstacles (np.prod, Domain.py:384)
        return inDom * np.logical_not(np.prod(inObs, axis=1))

    def innerDisc(self, discNum, rfrac=0., sortflg=True, discTol=None):
        if np.size(discNum) not in (1, 2):
            raise ValueError('\'discNum\' dimension incompatible!')
        if np.size(discNum) == 1:
            d = int(np.reshape(discNum, -1)[0])
            discNum = [d, d]
        if discTol is not None and np.size(discTol) not in (1, 2):
            raise ValueError('\'discTol\' dimension incompatible!')
        if discTol is not None and np.size(discTol) == 1:
            discTol = [float(np.reshape(discTol, -1)[0])] * 2
        lim = self.lim
        rf = rfrac ** 0.5
        axes, he = [], []
        for d in range(2):
            dof = int(discNum[d])
            h = (lim[1, d] - lim[0, d]) / (dof + 1)
            he.append(h)
            tol = h if discTol is None else float(np.reshape(discTol[d], -1)[0])
            axes.append(_axis_nodes(lim[0, d], lim[1, d], dof, rf, sortflg, tol))
        he = np.array(he)
        X, Y = np.meshgrid(axes[0], axes[1], indexing='xy')       # x fastest
        coord = np.stack([X.reshape(-1), Y.reshape(-1)], axis=1)
        return he, coord[self.isInside(coord), :]

    def boundaryDisc(self, vertices, bDiscNum, rfrac=0, sortflg=True):
        v = np.asarray(vertices, dtype=float)
        nxt = np.roll(v, -1, axis=0)
        edge = nxt - v
        length = np.linalg.norm(edge, axis=1)
        bdof, coord = [], []
        for i in range(v.shape[0]):
            n = math.ceil(bDiscNum * length[i])
            bdof.append(n)
            dof1 = math.floor(n * rfrac)
            step = np.hstack([np.random.uniform(size=dof1), np.linspace(0.0, 1.0, num=n - dof1)])
            if rfrac > 0 and sortflg:
                step = np.sort(step)
            coord.append(v[i, :] + edge[i, :] * step[:, None])
        return bdof, coord

    def getMesh(self, discNum=100, bDiscNum=50, rfrac=0, sortflg=True, discTol=None):
        he, coordinates = self.innerDisc(discNum, rfrac, sortflg, discTol)
        bdof, bCoordinates = self.boundaryDisc(self.vertices, bDiscNum, rfrac, sortflg)
        bIndNum = self.vertices.shape[0]
        for obs in self.obsVertices:
            bIndNum += np.shape(obs)[0]
            bd, bc = self.boundaryDisc(obs, bDiscNum, rfrac, sortflg)
            bdof.extend(bd)
            bCoordinates.extend(bc)
        return Mesh(dim=2, dof=coordinates.shape[0], coordinates=coordinates, he=he,
                    bIndNum=bIndNum, bdof=bdof, bCoordinates=bCoordinates, discNum=discNum,
                    bDiscNum=bDiscNum)


class Domain1D(Domain):
    """Interval domain (Domain.py:586-742)."""

    def __init__(self, interval=np.array([-1.0, 1.0])):
        interval = np.asarray(interval, dtype=float)
        if interval.ndim != 1:
            raise ValueError('interval must be a vector!')
        super().__init__(1, np.reshape(interval, [2, 1]))
        self.bIndNum = 2
        self.measure = interval[1] - interval[0]

    def isInside(self, x, tol=0.):
        x = np.asarray(x)
        if x.shape[1] != self.dim:
            raise ValueError('Vertex dimensions are incompatible with domain dimension!')
        return (self.lim[0] + tol <= x) * (x <= self.lim[1] - tol)

    def getMesh(self, discNum=100, bDiscNum=None, rfrac=0, sortflg=True, discTol=None):
        if np.size(discNum) != 1:
            raise ValueError('number of discretization points must be a scalar!')
        discNum = int(np.reshape(discNum, -1)[0])
        rfrac = min(max(rfrac, 0), 1)
        lim = self.lim
        he = (lim[1] - lim[0]) / (discNum + 1)                      # shape (1,), as the reference
        tol = float(he[0]) if discTol is None else float(np.reshape(discTol, -1)[0])
        c = _axis_nodes(float(lim[0, 0]), float(lim[1, 0]), discNum, rfrac, sortflg, tol)
        coordinates = np.reshape(c, [discNum, 1])
        bdof = np.ones(2, dtype=int)
        bCoordinates = np.reshape(lim, [2, 1, 1])
        return Mesh(dim=1, dof=discNum, coordinates=coordinates, he=he, bIndNum=2, bdof=bdof,
                    bCoordinates=bCoordinates, discNum=discNum)
