"""
varnet_amd -- MI355X-native engine for VarNet's variational-loss training loop behind the
reference's VarNet / ADPDE / Domain1D / PolygonDomain2D / MOR constructor API.

    from varnet_amd import Domain1D, ADPDE, VarNet
    pde = ADPDE(Domain1D(), diff=0.1/np.pi, vel=1.0, tInterval=[0, 2.0], IC=lambda x: -np.sin(np.pi*x))
    vn  = VarNet(pde, layerWidth=[20], discNum=20, bDiscNum=None, tDiscNum=300)
    vn.train('out', weight=[10., 10., 1.], epochNum=1000)

The device work runs in hand-written gfx950 HIP kernels reached through the C ABI of
libvarnet_hip.so (include/varnet_hip.h); there is no CPU fallback.
"""
__version__ = '0.1.0'

from .utility import UF
from .finite_element import FE
from .domain import Domain, Domain1D, PolygonDomain2D, Mesh
from .adpde import ADPDE
from .mor import MOR
from .varnet import VarNet, FIXData, ManageTrainData, TrainResult
from .contour import ContourPlot

__all__ = ['UF', 'FE', 'Domain', 'Domain1D', 'PolygonDomain2D', 'Mesh', 'ADPDE', 'MOR', 'VarNet',
           'FIXData', 'ManageTrainData', 'TrainResult', 'ContourPlot']
