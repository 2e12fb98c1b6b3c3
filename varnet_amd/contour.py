"""
`ContourPlot` -- the sampling grid and the field arrays behind the reference's result plots
(/root/reference/ContourPlot.py:55-296), without the drawing: `field()` returns exactly the
[discNum, discNum] array `conPlot` hands to `plt.contourf` (points outside the domain set to `fill_val`),
`snap()` the curve `snap1Dt` plots.  `conPlot` / `snap1Dt` keep the reference signatures and draw only when
matplotlib is importable and `draw=True`; the arrays are returned either way, so post-processing does not
need a display (VarNet.simRes uses them).
"""
import numpy as np

from .utility import UF

uf = UF()


class ContourPlot:
    def __init__(self, domain, tInterval=None, discNum=51):
        dim, lim = domain.dim, domain.lim
        hx = (lim[1, 0] - lim[0, 0]) / (discNum - 1)
        x_coord = np.linspace(lim[0, 0], lim[1, 0], discNum)
        if dim == 1 and uf.isnone(tInterval):
            raise ValueError('contour plot unavailable for 1D, time-independent problems!')
        elif dim == 1:
            status = '1D-time'
            hy = (tInterval[1] - tInterval[0]) / (discNum - 1)
            y_coord = np.linspace(tInterval[0], tInterval[1], discNum)
        if dim == 2:
            hy = (lim[1, 1] - lim[0, 1]) / (discNum - 1)
            y_coord = np.linspace(lim[0, 1], lim[1, 1], discNum)
            status = '2D' if uf.isnone(tInterval) else '2D-time'
        if dim > 2:
            raise ValueError('contour plots are available for 1D and 2D domains!')
        xx, yy = np.meshgrid(x_coord, y_coord, sparse=False)
        X_coord = np.tile(x_coord, discNum).reshape(-1, 1)               # x fastest
        Y_coord = np.repeat(y_coord, discNum).reshape(-1, 1)
        if status == '1D-time':
            isOutside = np.zeros(discNum ** 2, dtype=bool)
        else:
            isOutside = np.logical_not(domain.isInside(np.concatenate([X_coord, Y_coord], axis=1)))
        self.status, self.discNum, self.tInterval = status, discNum, tInterval
        self.he = np.array([hx, hy])
        self.isOutside = isOutside
        self.x_coord, self.y_coord = x_coord.reshape(discNum, 1), y_coord.reshape(discNum, 1)
        self.X_coord, self.Y_coord = X_coord, Y_coord
        self.xx, self.yy = xx, yy
        self.domain = domain

    # -- arrays ---------------------------------------------------------------------------------------
    def field(self, func, t=None, fill_val=0.):
        """The [discNum, discNum] array `conPlot` draws (ContourPlot.py:150-173)."""
        if not callable(func):
            raise ValueError('field function must be callable!')
        if self.status == '2D-time' and uf.isnone(t):
            raise ValueError('time must be provided for 2D time-dependent problems!')
        n = self.discNum
        if self.status == '1D-time':
            f = func(self.X_coord, self.Y_coord)
        elif self.status == '2D':
            f = func(np.concatenate([self.X_coord, self.Y_coord], axis=1))
        else:
            f = func(np.concatenate([self.X_coord, self.Y_coord], axis=1), t)
        f = np.array(f, dtype=float)
        if f.shape[0] != n ** 2:
            raise ValueError('output of the function should be a column vector with size {}!'.format(n ** 2))
        f = f.reshape(n ** 2, 1)
        f[self.isOutside, :] = fill_val
        return f.reshape(n, n)

    def snap(self, func, t):
        """(x_coord, func(x_coord, t)): what `snap1Dt` plots (ContourPlot.py:278-279)."""
        if not callable(func):
            raise ValueError('field function must be callable!')
        if self.status != '1D-time':
            raise ValueError('Function is specific to 1D time-dependent problems!')
        return self.x_coord, func(self.x_coord, t)

    # -- optional drawing -----------------------------------------------------------------------------
    def conPlot(self, func, t=None, figNum=None, title=None, fill_val=0., draw=True):
        field = self.field(func, t, fill_val)
        if draw:
            try:
                import matplotlib.pyplot as plt
            except ImportError:
                return field
            plt.figure(0 if figNum is None else figNum)
            cP = plt.contourf(self.xx, self.yy, field)
            plt.colorbar(cP)
            plt.xlabel('$x$' if self.status == '1D-time' else '$x_1$')
            plt.ylabel('time' if self.status == '1D-time' else '$x_2$')
            if title is not None:
                plt.title(title)
            plt.axis('scaled')
        return field

    def animPlot(self, func, t=[], figNum=None, title=None, fill_val=0., draw=True):
        """Frames of a 2D time-dependent field (ContourPlot.py:199-258): returns the list of [discNum, discNum] fields for
        the times `t` (default: 5 over the time interval); draws them one after the other when matplotlib is there."""
        if not callable(func):
            raise ValueError('field function must be callable!')
        if self.status in ('1D-time', '2D'):
            raise ValueError('animation contour plot is only available for 2D time-dependent problems!')
        if np.size(t) == 0:
            t = np.linspace(self.tInterval[0], self.tInterval[1], num=5)
        frames = []
        for ti in t:
            f = self.field(func, float(ti), fill_val)
            frames.append(f)
            if draw:
                try:
                    import matplotlib.pyplot as plt
                except ImportError:
                    continue
                plt.figure(0 if figNum is None else figNum)
                cP = plt.contourf(self.xx, self.yy, f)
                plt.colorbar(cP)
                tt = 't = {0:.2f}s'.format(ti)
                plt.title(tt if title is None else title + '-' + tt)
                plt.xlabel('$x_1$')
                plt.ylabel('$x_2$')
                plt.axis('scaled')
        return frames

    def snap1Dt(self, func, t, lineOpt=None, figNum=None, title=None, draw=True):
        x, f = self.snap(func, t)
        if draw:
            try:
                import matplotlib.pyplot as plt
            except ImportError:
                return f
            plt.figure() if figNum is None else plt.figure(figNum)
            plt.plot(x, f) if lineOpt is None else plt.plot(x, f, lineOpt)
            plt.xlabel('$x$')
            if title is not None:
                plt.title(title)
            plt.grid(True)
        return f
