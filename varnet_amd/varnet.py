"""
`VarNet` -- trainer / data assembly for the variational (weak-form) loss, with the reference's
constructor and `train / evaluate / residual / loadModel` surface
(/root/reference/VarNet.py:67-203, 1197-1692), re-designed around a GPU-resident engine:

  reference (TF1)                                   here (MI355X)
  ------------------------------------------------  ------------------------------------------
  nT-row fp64 NumPy arrays re-fed host->device on   built once on the host in fp64 (same
  EVERY step through sess.run(feed_dict)            arithmetic, so the fp32 values the device
  (VarNetUtility.py:840-854, 1044)                  sees are identical), uploaded once, resident
  N / dNt tiled to nT rows (FiniteElement.py:426)   period-integNum tables inside the engine
  towers in one process, grads summed on a          one process per GPU, contiguous test-function
  controller (TFModel.py:342-377)                   shard per rank, SUM all-reduce of the flat
                                                    gradient over RCCL (torch.distributed)
  loss read back every step (VarNetUtility.py:1044) accumulated on device, read once per epoch

Out of scope here: modelId='RNN', updateWeights, plots.
"""
import math
import os
import pickle
from datetime import datetime
import time
import warnings

import numpy as np

from .finite_element import FE
from .utility import UF

uf = UF()
shape = np.shape
size = np.size

# `train(dedup='auto')`: time model of one step in either formulation, fitted to 36 measured (grid, net) pairs on one MI355X
# (tools/dedup_auto_perf.py -> profiles/r6_dedup_auto_perf.txt; every pair the model sends to the de-duplicated formulation was
# measured faster there, and no pair it keeps row-wise was more than 4 % faster de-duplicated).  S = sum of the hidden widths:
#   row-wise        rows * row_ps * S
#   de-duplicated   unique points * point_ps * S + rows * asm_ps (the two assembly kernels) + fixed_us + param_ns * P
#                   (three more kernel boundaries and a second prologue that images the P parameters into LDS)
DEDUP_MODEL = {'row_ps': 4.5, 'point_ps': 6.2, 'asm_ps': 10.0, 'fixed_us': 20.0, 'param_ns': 2.2}


def unique_points(Input, feDim, hVec, device=None):
    """
    Unique quadrature points of an `Input` block (rows = test function x quadrature point).
    On the uniform grid the 2^feDim hat functions around an element share its quadrature points
    (VarNet.py:576-588), so rows repeat; coordinates reached from different nodes differ only by
    fp64 rounding and are merged on a lattice of h/4096.  Returns (first-occurrence row of every
    unique point, uid [n] row -> unique point, rowptr [U+1], rowidx [n] CSR inverse).
    With a CUDA `device` the sorts run there (torch.unique / stable argsort: the same maps, bit for bit, as the NumPy path --
    unique keys ascending, first occurrences, rows of a point in increasing order; 1.4 s -> 0.1 s at 15.4 M rows).
    """
    X = np.asarray(Input)[:, :feDim]
    h = np.reshape(np.asarray(hVec, dtype=float), (1, feDim))
    if device is not None and getattr(device, 'type', None) == 'cuda':
        import torch
        Xd = torch.as_tensor(np.ascontiguousarray(X, dtype=np.float64), device=device)
        k = torch.round((Xd - Xd.min(dim=0, keepdim=True).values) / torch.as_tensor(h, device=device) * 4096.0).to(torch.int64)
        kmax = k.max(dim=0).values.tolist()
        bits = [int(np.ceil(np.log2(max(int(m), 1) + 1))) for m in kmax]
        if sum(bits) <= 62:
            key = torch.zeros(k.shape[0], dtype=torch.int64, device=device)
            for d in range(feDim):
                key = (key << bits[d]) | k[:, d]
            _, uid = torch.unique(key, sorted=True, return_inverse=True)
            U = int(uid.max().item()) + 1
            n = uid.shape[0]
            first = torch.full((U,), n, dtype=torch.int64, device=device).scatter_reduce_(
                0, uid, torch.arange(n, dtype=torch.int64, device=device), reduce='amin', include_self=True)
            rowidx = torch.argsort(uid, stable=True)
            rowptr = torch.zeros(U + 1, dtype=torch.int64, device=device)
            rowptr[1:] = torch.cumsum(torch.bincount(uid, minlength=U), 0)
            return (first.cpu().numpy(), uid.to(torch.int32).cpu().numpy(), rowptr.to(torch.int32).cpu().numpy(),
                    rowidx.to(torch.int32).cpu().numpy())
    k = np.rint((X - X.min(axis=0, keepdims=True)) / h * 4096.0).astype(np.int64)
    bits = [int(np.ceil(np.log2(max(int(k[:, d].max()), 1) + 1))) for d in range(feDim)]
    if sum(bits) <= 62:
        key = np.zeros(X.shape[0], dtype=np.int64)
        for d in range(feDim):
            key = (key << bits[d]) | k[:, d]
        _, first, uid = np.unique(key, return_index=True, return_inverse=True)
    else:
        _, first, uid = np.unique(k, axis=0, return_index=True, return_inverse=True)
    uid = np.reshape(uid, -1).astype(np.int32)
    U = first.shape[0]
    rowidx = np.argsort(uid, kind='stable').astype(np.int32)
    rowptr = np.zeros(U + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(np.bincount(uid, minlength=U))
    return first, uid, rowptr, rowidx


# ======================================================================================
class FIXData:
    """
    Static discretisation data (/root/reference/VarNetUtility.py:204-463).  Differences from
    the reference: `N`, `dNx`, `dNt` are the period-`integNum` tables ([integNum], [integNum,dim],
    [integNum]) instead of nT-row tilings -- `rows()` materialises the tiled form on demand.
    """

    def __init__(self, vn, integPnum=2):
        dim = vn.dim
        PDE = vn.PDE
        domain = PDE.domain
        MORvar = PDE.MORvar
        timeDependent = PDE.timeDependent
        if timeDependent:
            feDim = dim + 1
            tDiscNum = vn.tDiscNum
            ht, t_coord = vn.timeDisc()
        else:
            feDim = dim
            tDiscNum = 1
            t_coord = []
        mesh = domain.getMesh(vn.discNum, vn.bDiscNum)
        dof = mesh.dof
        he = np.reshape(mesh.he, [dim, 1])
        hVec = np.vstack([he, ht]) if timeDependent else he
        uniform_biInput, biDof = vn.biTrainPoints(mesh, t_coord)
        nt = dof * tDiscNum
        biDimVal = domain.measure                                   # VarNetUtility.py:293

        lossVecflag = True
        if nt > 1e6:                                                # VarNetUtility.py:300-303
            lossVecflag = False
            mesh = domain.getMesh(discNum=100, bDiscNum=50)
            if timeDependent:
                _, t_coord = vn.timeDisc(tdof=100)
        coord = mesh.coordinates
        uniform_input = uf.pairMats(coord, t_coord) if timeDependent else coord
        if nt > 1e6:
            uniform_biInput, _ = vn.biTrainPoints(mesh, t_coord)

        if MORvar is not None:
            discArg = MORvar.discretizeArg(vn.MORdiscScheme)
            argInd = MORvar.argIndex(discArg)
            batchNum = len(argInd)
            if batchNum > 16:
                lossVecflag = False
        else:
            discArg, argInd, batchNum = None, None, 1

        self.dim, self.feDim, self.timeDependent = dim, feDim, timeDependent
        self.integPnum = integPnum
        self.dof, self.bdof = dof, mesh.bdof
        self.biDof0, self.nt0 = biDof, nt
        self.hVec, self.biDimVal = hVec, biDimVal
        self.detJvec = False
        self.lossVecflag = lossVecflag
        self.uniform_input, self.uniform_biInput = uniform_input, uniform_biInput
        self.MORbatchNum, self.MORargInd, self.MORdiscArg = batchNum, argInd, discArg
        self.cEx = self.uniform_inpData = self.d_diff = None
        self.integNum = self.biDof = self.bDofsum = self.nt = self.nT = None
        self.delta = self.integW = self.detJ = self.N = self.dNx = self.dNt = None

    def setInputData(self, vn):
        """Exact solution, PDE fields and grad(kappa) on `uniform_input` (VarNetUtility.py:364-411)."""
        dim, PDE = self.dim, vn.PDE
        ui = self.uniform_input
        Coord = ui[:, :dim]
        targ = [ui[:, dim:dim + 1]] if self.timeDependent else []
        self.cEx = PDE.cEx(Coord, *targ) if PDE.cEx is not None else None
        self.uniform_inpData = vn.PDEinpData(ui) if PDE.MORvar is None else [None] * 3
        self.d_diff = PDE.d_diffFun(Coord, *targ)

    def setFEdata(self):
        """FE tables of the initial uniform sampling (VarNetUtility.py:415-462)."""
        biDof = self.biDof0
        self.bDofsum = int(np.sum(biDof[:-1])) if self.timeDependent else int(np.sum(biDof))
        fe = FE(self.feDim, self.integPnum)
        integNum, detJ, delta, integW, N, dN = fe.basisTable(self.hVec)
        self.integNum, self.biDof = integNum, biDof
        self.segments, self.detJvec = None, False
        self.nt, self.nT = self.nt0, self.nt0 * integNum
        self.delta, self.integW, self.detJ = delta, integW, detJ
        self.N = N
        self.dNx = dN[:, 0:self.dim]
        self.dNt = dN[:, self.dim] if self.timeDependent else np.zeros(integNum)

    def rows(self, nt=None):
        """Tiled [nT,1], [nT,dim], [nT,1] forms of N, dNx, dNt (what the reference stores).  After
        `updateOptimData` with a support scaling the rows of the added (optimal) test functions
        come first and use their own tables (VarNetUtility.py:506-523)."""
        if getattr(self, 'segments', None):
            Ns, dNxs, dNts = [], [], []
            for cnt, N, dNx, dNt in self.segments:
                Ns.append(np.tile(N.reshape(-1, 1), (cnt, 1)))
                dNxs.append(np.tile(dNx, (cnt, 1)))
                dNts.append(np.tile(dNt.reshape(-1, 1), (cnt, 1)))
            return np.vstack(Ns), np.vstack(dNxs), np.vstack(dNts)
        nt = self.nt if nt is None else nt
        return (np.tile(self.N.reshape(-1, 1), (nt, 1)), np.tile(self.dNx, (nt, 1)),
                np.tile(self.dNt.reshape(-1, 1), (nt, 1)))

    def updateOptimData(self, frac, suppFactor):
        """
        Fixed data after `ceil(frac*nt0)` residual-driven test functions were ADDED in front of
        the uniform ones (VarNetUtility.py:466-545): new nt/nT/biDof/bDofsum; with a support scaling
        (`suppFactor != 1`) the added test functions get their own element sizes, so `detJ`
        becomes a per-test-function vector (`detJvec=True`) and N/dN per-row.
        """
        if self.nt > self.nt0:
            return
        nt0 = self.nt0
        nt1 = math.ceil(frac * nt0)
        scaled = np.abs(suppFactor - 1.0) > 1.e-15
        self.nt = nt0 + nt1
        self.nT = self.nt * self.integNum
        if scaled:
            fe = FE(self.feDim, self.integPnum)
            _, detJ1, _, _, N1, dN1 = fe.basisTable(suppFactor * self.hVec)
            dNt1 = dN1[:, self.dim] if self.timeDependent else np.zeros(self.integNum)
            self.segments = [(nt1, N1, dN1[:, 0:self.dim], dNt1), (nt0, self.N, self.dNx, self.dNt)]
            self.detJ = np.vstack([detJ1 * np.ones([nt1, 1]), self.detJ * np.ones([nt0, 1])])
            self.detJvec = True
        biDof = [b + math.ceil(frac * b) for b in self.biDof0]
        self.biDof = biDof
        self.bDofsum = int(np.sum(biDof[:-1])) if self.timeDependent else int(np.sum(biDof))


# ======================================================================================
class TrainResult:
    """
    Training record on disk with the reference's file formats (/root/reference/VarNetUtility.py:1149-1631),
    so existing post-processing keeps working:

      * `<folder>/caseData.txt` -- banner, date, problem header, the sections "Boundary condition
        information", "Neural Network architecture", "Processor information", "Optimizer information",
        "Space-time discretization information", "Sampling scheme ...", "Batch-optimization
        information", "Weighting information", "Stopping criteria", then "Training iterations:" followed
        by one "Epoch %2d: loss = %2.5f" line per reported epoch (:1217-1464, :1574-1588);
      * `<folder>/trainData.vn` -- pickle (HIGHEST_PROTOCOL) of the instance `__dict__` with the keys
        `casepath, plotpath, caseSimline, saveFreq, verbose, pltReplace, trainWeight, loss, lossComp,
        avgtime0, avgtime, residual, iterSmp, inpIter, error, lossVec, option_stopping,
        option_trainPoint, option_weighting, option_batchOptim` (:1512-1556);
      * histories are sampled every `saveFreq` epochs (`iterSmp`), epochs before the first `saveFreq` are
        only logged (:1577-1583).
    Plots (`iterPlot`) are out of scope; the `plots` folder is still created.
    """

    def __init__(self, folderpath, cExFlg=False, verbose=True, saveFreq=100, pltReplace=True):
        self.folderpath = folderpath
        self.verbose = verbose
        self.saveFreq = saveFreq
        self.pltReplace = pltReplace
        self.trainWeight = None
        self.casepath = None
        self.plotpath = None
        self.caseSimline = 0
        if folderpath is not None:
            self.plotpath = os.path.join(folderpath, 'plots')
            os.makedirs(self.plotpath, exist_ok=True)
            self.casepath = os.path.join(folderpath, 'caseData.txt')
        self.loss = []                    # total loss at the sampled epochs
        self.lossAll = []                 # (extension) total loss of every epoch
        self.lossComp = []                # [BC, IC, variational] at the sampled epochs
        self.avgtime0 = None              # reference iteration time from the first saveFreq epochs
        self.avgtime = 0
        self.residual = []
        self.iterSmp = []
        self.inpIter = []
        self.error = [] if not uf.isnone(cExFlg) else None      # (always a list: callers pass a bool)
        self.lossVec = None
        self.option_stopping = self.option_trainPoint = self.option_weighting = self.option_batchOptim = None

    # -- case file ------------------------------------------------------------------------------------
    def initializeCase(self, varNet, trainArg):
        g = trainArg.get
        self.option_stopping = {'epochNum': g('epochNum'), 'tol': g('tol')}
        self.option_trainPoint = {k: g(k) for k in ('smpScheme', 'frac', 'addTrainPts', 'suppFactor', 'multiTrainUpd',
                                                    'trainUpdelay', 'tolUpd', 'reinitrain')}
        self.option_weighting = {k: g(k) for k in ('weight', 'updateWeights', 'normalizeW', 'adjustWeight', 'useOriginalW')}
        self.option_batchOptim = {k: g(k) for k in ('saveMORdata', 'batchNum', 'batchLen', 'shuffleData', 'shuffleFreq')}
        if self.folderpath is None:
            return
        PDE, fd = varNet.PDE, varNet.fixData
        td, dim = PDE.timeDependent, varNet.dim
        MORoff = uf.isnone(PDE.MORvar)
        smp = g('smpScheme')
        bar = '-' * 79 + '\n'
        L = [bar, '=' * 31 + ' VarNet Library ' + '=' * 32 + '\n', bar,
             'MI355X-native engine (varnet_amd); file layout of the VarNet library, arXiv:1912.07443\n\n', bar]
        d, t = datetime.now().strftime('%d/%m/%Y %H:%M:%S').split()
        L.append('Simulation date: ' + d + ' - time: ' + t + '\n\n')
        L.append('%dD %s Advection-Diffusion problem %s model-order-reduction.\n\n'
                 % (dim, 'time-dependent' if td else 'steady-state', 'without' if MORoff else 'with'))
        L.append('Boundary condition information:\n')
        if dim == 1:
            L.append('\ttype:' + PDE.BCtype[0] + ', ' + PDE.BCtype[1] + '\n')
        else:
            geom = PDE.domain.boundryGeom.tolist()
            for bi in range(PDE.domain.bIndNum):
                L.append('\tBC%d: %s - vertices: %s\n' % (bi + 1, PDE.BCtype[bi], geom[bi]))
        L.append('\n')
        L.append('Neural Network architecture:\n')
        L.append('\ttype: ' + str(varNet.modelId) + '\n')
        L.append('\tnumber of inputs: ' + str(varNet.inpDim) + '\n')
        L.append('\tnumber of layers: ' + str(len(varNet.layerWidth)) + '\n')
        L.append('\tnumber of nodes in each layer: ' + str(varNet.layerWidth) + '\n')
        L.append('\tactivation function for each layer: ' + str(varNet.activationFun) + '\n')
        L.append('\ttotal number of trainable parameters: ' + str(varNet.engine.P) + '\n\n')
        L.append('Processor information:\n')
        if varNet.world > 1:
            L.append('\tparallel replicated training on %d processors\n' % varNet.world)
            L.append('\tutilized processors: ' + ' and '.join('GPU:%d' % r for r in range(varNet.world)) + '\n')
            L.append('master controller:GPU:0\n\n')
        else:
            L.append('\tutilized processor: GPU:0\n\n')
        L.append('Optimizer information:\n')
        L.append('\ttype: %s stochastic gradient descent algorithm\n'
                 % ('RMSProp' if str(varNet.optimizer).lower() == 'rmsprop' else 'Adam'))
        L.append('\tlearning rate: ' + str(varNet.learning_rate) + '\n\n')
        L.append('Space-time discretization information:\n')
        L.append('\tspatial domain interior discretization number: ' + str(varNet.discNum) + '\n')
        L.append('\tspatial domain boundary discretization density: ' + str(varNet.bDiscNum) + '\n')
        if td:
            L.append('\ttemporal discretization number: ' + str(varNet.tDiscNum) + '\n')
        L.append('\tnumber of training points: ' + str(fd.nt) + '\n')
        L.append('\tnumber of training points for BCs: ' + str(list(fd.biDof)[:-1]) + '\n')
        L.append('\ttotal number of BC training points: ' + str(fd.bDofsum) + '\n')
        if td:
            L.append('\tnumber of training points for IC: ' + str(list(fd.biDof)[-1]) + '\n')
        L.append('\n')
        L.append('Sampling scheme for training points: ' + str(smp) + '\n')
        if smp != 'uniform':
            L.append('\tfraction of non-uniform points: ' + str(g('frac')) + '\n')
            if g('addTrainPts'):
                L.append('\tnon-uniform points are added without replacement\n')
            if g('multiTrainUpd'):
                L.append('\tnon-uniform points updated every ' + str(g('trainUpdelay')) + ' epochs ...\n')
                L.append('\t... if decrease in 5 consecutive loss values is less than ' + str(g('tolUpd')) + '\n')
            else:
                L.append('\tnon-uniform points updated once after ' + str(g('trainUpdelay')) + ' epochs\n')
                L.append('\tif decrease in 5 consecutive loss values is less than ' + str(g('tolUpd')) + '\n')
            if smp == 'optimal' and np.abs(g('suppFactor') - 1.0) > 1.e-15:
                L.append('\tsupport of optimal training points is scaled by a factor of ' + str(g('suppFactor')) + '\n')
            if g('reinitrain'):
                L.append('\ttrainable variables are re-initialized after update of training points\n')
            L.append('Note: since for non-uniform grid the training points and possibly weights are updated\n'
                     '      the loss component plots will not necessarily match total loss plot!\n')
        L.append('\n')
        bN, bL = g('batchNum'), g('batchLen')
        if not (MORoff and uf.isnone(bN) and uf.isnone(bL)):
            L.append('Batch-optimization information:\n')
            if not MORoff:
                L.append('\tnumber of MOR batches: ' + str(fd.MORbatchNum) + '\n')
                if g('saveMORdata'):
                    L.append('\tMOR fields are stored for faster training.\n')
            if not uf.isnone(bN):
                L.append('\tnumber of training batches: ' + str(bN) + '\n')
            elif not uf.isnone(bL):
                L.append('\tlength of training batches: ' + str(bL) + '\n')
            if g('shuffleData'):
                L.append('\tshuffle training data every ' + str(g('shuffleFreq')) + ' epochs\n')
            if not (uf.isnone(bN) and uf.isnone(bL)):
                L.append('Note: for batch-optimization loss component values will not match total loss\n'
                         '      since intermediate iterations change total loss unlike loss components!\n')
            L.append('\n')
        L.append('Weighting information:\n')
        L.append('\trequested weights: ' + str(g('weight')) + '\n')
        if g('updateWeights'):
            L.append('\tweights updated to maintain the requested balance between terms\n')
        if g('normalizeW'):
            L.append('\tweights normalized by values so that weight times values have requested weights\n')
        if smp != 'uniform' and g('adjustWeight'):
            L.append('\tweights on boundary-initial conditions updated after addition of non-uniform points\n')
        if g('useOriginalW'):
            L.append('\trequested weights applied without any modification\n')
        L.append('\n')
        L.append('Stopping criteria:\n')
        L.append('\tmaximum number of epochs: ' + str(g('epochNum')) + '\n')
        L.append('\tstopping tolerance: ' + str(g('tol')) + '\n\n')
        L.append('=' * 58 + '\n')
        L.append('Training iterations:\n\n')
        with open(self.casepath, 'w') as f:
            f.write(''.join(L))
        with open(self.casepath) as f:
            self.caseSimline = len(f.readlines()) - 4      # comments are inserted above the iteration header

    def writeCase(self, text):
        if not isinstance(text, str):
            raise ValueError('the input must be a string!')
        if self.casepath is None:
            return
        with open(self.casepath, 'a+') as f:
            f.write(text)

    def writeComment(self, text):
        """Insert `text` above the training iterations (post-simulation notes at the top of the case file)."""
        if not isinstance(text, str):
            raise ValueError('the input must be a string!')
        if self.casepath is None:
            return
        lines = [ln + '\n' for ln in text.split('\n')]
        with open(self.casepath) as f:
            data = f.readlines()
        data[self.caseSimline:self.caseSimline] = lines
        with open(self.casepath, 'w') as f:
            f.writelines(data)

    # -- pickle ----------------------------------------------------------------------------------------
    def saveData(self):
        if self.folderpath is None:
            return
        with open(os.path.join(self.folderpath, 'trainData.vn'), 'wb') as f:
            pickle.dump(self.__dict__, f, pickle.HIGHEST_PROTOCOL)

    def loadData(self):
        with open(os.path.join(self.folderpath, 'trainData.vn'), 'rb') as f:
            dump = pickle.load(f)
        for key in ('casepath', 'plotpath', 'caseSimline', 'saveFreq', 'verbose', 'pltReplace', 'trainWeight', 'loss',
                    'lossComp', 'avgtime0', 'avgtime', 'residual', 'iterSmp', 'inpIter', 'error', 'lossVec'):
            setattr(self, key, dump[key])
        try:                                                   # files written before the options were stored
            for key in ('option_stopping', 'option_trainPoint', 'option_weighting', 'option_batchOptim'):
                setattr(self, key, dump[key])
        except KeyError:
            warnings.warn('train() arguments not loaded!')

    # -- per-epoch report ------------------------------------------------------------------------------
    def iterOutput(self, epoch, current_loss, min_loss, epoch_time, resVal, err, lossSplit, lossVec):
        saveFreq = self.saveFreq
        self.lossAll.append(current_loss)
        if not (epoch < saveFreq or epoch % saveFreq == 0):
            return
        line = 'Epoch %2d: loss = %2.5f\n' % (epoch, current_loss)
        self.writeCase(line)
        if self.verbose:
            print(line, end='')
        if epoch < saveFreq:
            if epoch == saveFreq - 1:
                self.avgtime0 = epoch_time / epoch
            return
        self.iterSmp.append(epoch)
        self.loss.append(current_loss)
        self.lossComp.append(np.reshape(lossSplit, 3))
        self.residual.append(resVal)
        if self.error is not None:
            self.error.append(err)
        self.avgtime = epoch_time / epoch
        self.lossVec = lossVec
        self.saveData()
        if epoch % (10 * saveFreq) == 0:
            msg = '\nbest model loss: %2.5f\naverage iteration time: %2.5fs\n\n' % (min_loss, epoch_time / epoch)
            if self.verbose:
                print(msg, end='')
            self.writeCase(msg)
            # The reference redraws its five figures here (1.2 s at 300 dpi).  Its epochs take seconds; here 10 * saveFreq
            # epochs can pass in milliseconds, so a refresh is skipped while the last one is younger than `plotEvery`
            # seconds and VarNet.train flushes the pending one when it returns: same files, same final content.
            self._plot_dirty = True
            if time.perf_counter() - getattr(self, '_plot_last', -1e30) >= getattr(self, 'plotEvery', 30.0):
                self.iterPlot()

    def flushPlots(self):
        if getattr(self, '_plot_dirty', False):
            self.iterPlot()

    def iterPlot(self, plotpath=None, pltFrmt='png'):
        """
        Convergence plots the reference refreshes every 10 * saveFreq epochs (VarNetUtility.py:1634-1756), same file
        names: `loss`, `lossComp`, `scaled_lossComp`, `res_history` and, with an exact solution, `error` (prefixed by
        the epoch when pltReplace is False); dashed red lines mark the epochs at which the training set was redrawn.
        """
        if pltFrmt not in ('png', 'jpg', 'pdf', 'eps'):
            raise ValueError('invalid plot format!')
        self._plot_dirty = False
        self._plot_last = time.perf_counter()
        if self.folderpath is None or not self.iterSmp:          # a tower other than rank 0 keeps no files
            return
        try:
            import matplotlib
            matplotlib.use('Agg', force=False)
            import matplotlib.pyplot as plt
        except ImportError:                                   # no plotting library: the records in trainData.vn remain
            return
        ext = '.' + pltFrmt
        plotpath = self.plotpath if plotpath is None else plotpath
        os.makedirs(plotpath, exist_ok=True)
        pre = '' if self.pltReplace else '{}_'.format(self.iterSmp[-1])
        png = pltFrmt == 'png'
        it = np.asarray(self.iterSmp, dtype=float)
        it0 = np.concatenate([[0.0], it])
        comp = np.array([np.reshape(c, -1) for c in self.lossComp], dtype=float)

        def finish(fig, name, ylabel, title, legend=None):
            ax = fig.gca()
            ax.set_xlabel('epochs')
            ax.set_ylabel(ylabel)
            ax.set_title(title if png else '')
            if legend:
                ax.legend(legend)
            ax.grid(True)
            fig.savefig(os.path.join(plotpath, pre + name + ext), dpi=300)
            plt.close(fig)

        fig = plt.figure()
        fig.gca().semilogy(it, np.asarray(self.loss, dtype=float))
        for i in (self.inpIter or []):
            fig.gca().axvline(i, color='r', linestyle='--')
        finish(fig, 'loss', 'loss function', 'convergence plot (variable grid)')
        fig = plt.figure()
        for col, c in zip(range(3), 'brg'):
            fig.gca().semilogy(it0, comp[:, col], c)
        finish(fig, 'lossComp', 'loss components', 'loss components plot', ['BC', 'IC', 'integral term'])
        fig = plt.figure()
        tot = np.asarray(self.trainWeight, dtype=float) * comp
        for col, c in zip(range(3), 'brg'):
            fig.gca().semilogy(it0, tot[:, col], c)
        fig.gca().semilogy(it0, np.sum(tot, axis=1), 'k')
        finish(fig, 'scaled_lossComp', 'loss components', 'scaled loss components plot', ['BC', 'IC', 'integral term', 'total loss'])
        fig = plt.figure()
        fig.gca().semilogy(it, np.asarray(self.residual, dtype=float))
        finish(fig, 'res_history', 'residual', 'residual convergence plot')
        if self.error is not None:
            fig = plt.figure()
            fig.gca().semilogy(it, np.asarray(self.error, dtype=float))
            finish(fig, 'error', 'error', 'normalized solution error')


# ======================================================================================
class ManageTrainData:
    """
    Device-resident training set, sharded and batched the way the reference's feed dicts are
    (/root/reference/VarNetUtility.py:563-1017): whole test functions only, mini-batch x tower
    blocks of `batchLen = ceil(nt/batchNum/puNum)` consecutive test functions, BC/IC set
    replicated with its weights divided by batchNum*puNum.
    """

    def __init__(self, vn, mor_data, batchNum=None, batchLen=None):
        if batchNum is not None and batchLen is not None:
            raise ValueError('Only one of batch number or length properties must be provided!')
        self.vn = vn
        self.mor = mor_data                      # list (per MOR batch) of dicts of device tensors
        fd = vn.fixData
        self.nt, self.integNum = fd.nt, fd.integNum
        # snapshot of the fixed data this set was built with (the trainer may later re-sample)
        self.detJ, self.bDofsum, self.biDimVal = fd.detJ, fd.bDofsum, fd.biDimVal
        puNum = vn.world
        if batchNum is None and batchLen is None:
            batchNum = 1
        if batchNum is None:
            batchLen = min(int(batchLen), self.nt)
            batchNum = int(np.ceil(self.nt / batchLen / puNum))
        else:
            batchNum = int(batchNum)
            batchLen = int(np.ceil(self.nt / batchNum / puNum))
        self.batchNum, self.batchLen, self.puNum = batchNum, batchLen, puNum
        self.batchInd = np.arange(self.nt)
        self.shuffled = False
        self.dedup_on, self.dedup_reason = False, None
        self._register()

    def block(self, bi, tower=None):
        """[n0,n1) test-function range of a tower (default: this rank's) in mini-batch bi."""
        j = bi * self.puNum + (self.vn.rank if tower is None else tower)
        n0 = min(j * self.batchLen, self.nt)
        return n0, min(n0 + self.batchLen, self.nt)

    def engine_batch(self, mor_b, bi):
        return mor_b * self.batchNum + bi

    def towerWeights(self, trainW):
        """Weights one tower is fed: BC/IC terms divided by batchNum*puNum because every (mini-batch, tower)
        feed repeats the whole BC/IC set (updateDictFields('trainW'), VarNetUtility.py:900-901)."""
        w = np.array(trainW, dtype=float)
        w[:-1] = w[:-1] / self.batchNum / self.puNum
        return w

    def _register(self):
        eng, q = self.vn.engine, self.integNum
        torch = eng.torch
        if self.shuffled:                         # one upload of the permutation per shuffle, not one per block
            perm_dev = self._upload_perm(torch, eng.device)
            rows_all = (perm_dev[:, None] * q + torch.arange(q, device=eng.device)[None, :]).reshape(-1)
        for mb, d in enumerate(self.mor):
            if self.shuffled:                     # one gather per array and parameter batch; the blocks are views of it
                d = dict(d)
                for key in ('Input', 'gcoef', 'source', 'N_rows', 'dNt_rows'):
                    if d.get(key) is not None:
                        d[key] = d[key].index_select(0, rows_all)
                if d.get('detJ') is not None:
                    d['detJ'] = d['detJ'].index_select(0, perm_dev)
            for bi in range(self.batchNum):
                n0, n1 = self.block(bi)
                pick = lambda t: None if t is None else t[n0 * q:n1 * q]
                pick_k = lambda t: None if t is None else t[n0:n1]
                detJ = pick_k(d.get('detJ'))
                eng.set_interior(self.engine_batch(mb, bi), pick(d['Input']), pick(d['gcoef']), pick(d['source']),
                                 n_k=n1 - n0, detJ=self.detJ if detJ is None else detJ,
                                 N_rows=pick(d.get('N_rows')), dNt_rows=pick(d.get('dNt_rows')))
                perm = getattr(self, 'biPerm', {}).get(bi)
                if perm is not None and hasattr(eng, 'set_batch_bic'):
                    ix = torch.as_tensor(perm, device=eng.device, dtype=torch.long)
                    eng.set_batch_bic(self.engine_batch(mb, bi), d['biInput'].index_select(0, ix),
                                      d['biLabel'].index_select(0, ix))

    def _upload_perm(self, torch, device):
        """Test-function permutation -> device.  On the GPU through one of two persistent pinned buffers and an
        asynchronous copy (a pageable upload would wait for the running epoch; allocating pinned memory per shuffle costs
        tens of ms); a buffer is rewritten only after the event behind its last copy has completed."""
        if device.type != 'cuda':
            return torch.as_tensor(self.batchInd, device=device, dtype=torch.long)
        if not hasattr(self, '_pin'):
            n = len(self.batchInd)
            self._pin = [torch.empty(n, dtype=torch.long).pin_memory() for _ in range(2)]
            self._pin_ev = [None, None]
            self._pin_i = 0
        i = self._pin_i = 1 - self._pin_i
        if self._pin_ev[i] is not None:
            self._pin_ev[i].synchronize()
        self._pin[i].numpy()[:] = self.batchInd
        out = self._pin[i].to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self._pin_ev[i] = ev
        return out

    def shuffleTrainData(self):
        """
        Reshuffle the mini-batches exactly like the reference (VarNetUtility.py:957-1017), drawing from the global NumPy
        stream in its call order, so that the same `np.random.seed` gives the same feeds: one permutation of the
        test-function order, then -- per (mini-batch, tower), cumulatively -- a permutation of ALL boundary/initial rows
        that the feed uses in place of the ordered set.  (That second permutation crosses the BC/IC split, so the rows
        that enter the BC mean and the IC mean change; with MOR it is overwritten by the next `trainData` call of every
        batch, VarNetUtility.py:921-926, and therefore has no effect there -- both reproduced.)  With towers, rank 0's
        seed is broadcast first so that every rank draws the same permutations.
        """
        self.vn._sync_sampling()
        np.random.shuffle(self.batchInd)
        nB = int(self.mor[0]['biInput'].shape[0])
        biInd = np.arange(nB)
        self.biPerm = {}
        for bi in range(self.batchNum):
            for tower in range(self.puNum):
                np.random.shuffle(biInd)
                if tower == self.vn.rank:
                    self.biPerm[bi] = biInd.copy()
        if self.vn.PDE.MORvar is not None:
            self.biPerm = {}                        # reset by updateDictFields before the batch is used
        self.shuffled = True
        self.dedup_on = False                       # vn_set_interior drops the registrations; a shuffled set is row-wise
        self._register()

    def activate(self):
        """(Re-)register this set's batches with the engine (after another set used it)."""
        self._register()
        if self.dedup_on:
            self.enable_dedup()

    def dedup_applies(self):
        """None when the de-duplicated formulation can serve this set, else the reason it cannot (a sentence)."""
        vn = self.vn
        fd = vn.fixData
        if self.shuffled:
            return 'the mini-batches are shuffled (the point map of every block would have to be rebuilt at each shuffle)'
        if fd.detJvec:
            return ('the training set is non-uniform (per-test-function supports, detJvec=True, VarNetUtility.py:466-545): '
                    'its rows share no quadrature points')
        if vn.dim > 3:
            return 'the problem has dim = %d > 3 space coordinates' % vn.dim
        if not hasattr(vn.engine, 'set_dedup'):
            return 'this engine has no vn_set_dedup'
        if not getattr(vn.engine, 'dedup_supported', lambda: True)():
            return ('the network %s with integNum = %d is outside the 8-wave fused kernel family the formulation runs on '
                    '(widths <= 64, <= 8 layers, integNum <= 256)' % (list(vn.layerWidth), self.integNum))
        return None

    def dedup_pays(self):
        """
        The rule behind `train(dedup='auto')`: True when the de-duplicated step is expected to be the faster one.
        It costs 8 F_pt per unique point in five launches against 6 F_pt per row in two, so it wins whenever the step is
        work-bound (up to 5.2 x on BASELINE config 3) and loses on steps so small that its three extra kernel boundaries and
        second prologue outweigh the work saved (0.5 x on an 8 000-row step).  Deterministic -- a function of the grid and
        the network, never of a timing -- so that a run is reproducible: DEDUP_MODEL, evaluated on the smallest block.
        """
        m = DEDUP_MODEL
        lw = [self.vn.inpDim] + list(self.vn.layerWidth) + [1]
        P = sum(a * b + b for a, b in zip(lw[:-1], lw[1:]))
        S = float(sum(self.vn.layerWidth))
        rows = min([(n1 - n0) * self.integNum for n0, n1 in map(self.block, range(self.batchNum)) if n1 > n0] or [0])
        if rows == 0:
            return False
        # the exact count of unique points is known only once the map is built (seconds of host work at 6.4 M rows): the
        # decision uses the grid's own ratio, 2^feDim rows per point reached from below
        U = rows / self.rows_per_point_estimate()
        t_row = rows * m['row_ps'] * S * 1e-6
        t_dd = U * m['point_ps'] * S * 1e-6 + rows * m['asm_ps'] * 1e-6 + m['fixed_us'] + m['param_ns'] * P * 1e-3
        return t_row >= t_dd

    def rows_per_point_estimate(self):
        """Rows per unique quadrature point of one block on the uniform grid, without building the map: per dimension a
        run of n consecutive hat functions covers n + 1 elements, so 2n element visits hit n + 1 distinct elements."""
        n0, n1 = self.block(0)
        nk = max(n1 - n0, 1)
        tdn = int(self.vn.tDiscNum) if self.vn.PDE.timeDependent else 1
        if self.vn.PDE.timeDependent:                 # blocks are contiguous in (space-major, time-minor) order
            nt_t = min(nk, tdn)
            ns = max(nk // tdn, 1)
        else:
            nt_t, ns = 1, nk
        r = 2.0 * nt_t / (nt_t + 1.0) if self.vn.PDE.timeDependent else 1.0
        per = max(ns ** (1.0 / self.vn.dim), 1.0)
        r *= (2.0 * per / (per + 1.0)) ** self.vn.dim
        return max(r, 1.0)

    def enable_dedup(self):
        """
        Switch every registered batch to the de-duplicated formulation (`vn_set_dedup`): one network
        evaluation per unique quadrature point instead of one per (test function, point) row.
        Needs the periodic FE tables (uniform supports) and an unshuffled set; returns the total
        number of unique points, or 0 if it does not apply -- `self.dedup_reason` then says why.
        """
        vn = self.vn
        fd = vn.fixData
        self.dedup_reason = self.dedup_applies()
        if self.dedup_reason is not None:
            return 0
        q, total = self.integNum, 0
        cache = getattr(self, '_dd_cache', {})
        for mb, d in enumerate(self.mor):
            for bi in range(self.batchNum):
                n0, n1 = self.block(bi)
                if n1 == n0:
                    continue
                key = (mb, bi)
                if key not in cache:
                    blk = d['Input_host'][n0 * q:n1 * q]
                    first, uid, rowptr, rowidx = unique_points(blk, fd.feDim, fd.hVec, getattr(vn.engine, 'device', None))
                    cache[key] = (vn.engine.dev(blk[first]), uid, rowptr, rowidx)
                Xu, uid, rowptr, rowidx = cache[key]
                # not applicable was decided above; an engine error here is a real failure and propagates
                vn.engine.set_dedup(self.engine_batch(mb, bi), Xu, uid, rowptr, rowidx)
                total += Xu.shape[0]
        self._dd_cache = cache
        self.dedup_on = True
        return total

    def disable_dedup(self):
        """Back to the row-wise formulation for every registered batch (`vn_set_dedup` with no points)."""
        if not self.dedup_on:
            return
        for mb in range(len(self.mor)):
            for bi in range(self.batchNum):
                n0, n1 = self.block(bi)
                if n1 > n0:
                    self.vn.engine.set_dedup(self.engine_batch(mb, bi))
        self.dedup_on = False

    def select_mor(self, mb):
        # (called once per epoch and MOR batch: skipped while the engine still holds THIS set's rows of batch mb -- any other
        # set_bic, from another training set or a direct caller, clears the engine's note)
        eng = self.vn.engine
        key = getattr(self, '_bic_token', None)
        if key is None:
            key = self._bic_token = object()
        if getattr(eng, '_bic_key', None) == (key, mb):
            return
        d = self.mor[mb]
        eng.set_bic(d['biInput'], d['biLabel'], self.bDofsum, self.biDimVal)
        eng._bic_key = (key, mb)


# ======================================================================================
class VarNet:
    def __init__(self, PDE, layerWidth=[20], modelId='MLP', activationFun=None, discNum=20,
                 bDiscNum=[], tDiscNum=[], MORdiscScheme=None, processors=None, controller=None,
                 integPnum=2, optimizer='adam', learning_rate=0.001):
        dim = PDE.dim
        timeDependent = PDE.timeDependent
        MORvar = PDE.MORvar
        # argument checks: /root/reference/VarNet.py:147-171
        if size(discNum) != 1 and size(discNum) != dim:
            raise ValueError('dimension of the number of discretizations does not match dimension of the domain!')
        elif size(discNum) == 1:
            discNum = [int(np.reshape(discNum, -1)[0])] * dim
        if size(bDiscNum) != 1:
            raise ValueError('density of boundary discretizations must be a scalar!')
        if modelId == 'RNN':
            raise NotImplementedError('modelId=\'RNN\' is unfinished in the reference (TFModel.py:225,763) '
                                      'and out of scope here')
        if modelId != 'MLP':
            raise ValueError('unknown modelId!')
        if timeDependent and uf.isempty(tDiscNum):
            raise ValueError('time discretization number must be provided for time-dependent PDEs!')
        if activationFun is None:
            activationFun = 'sigmoid'
        if not isinstance(layerWidth, list):
            raise ValueError('layer widths should be given in a list!')
        if MORvar is not None and MORdiscScheme is None:
            raise ValueError('\'MORdiscScheme\' must be given for MOR!')
        if learning_rate < 0.0:
            raise ValueError('learning rate must be positive!')
        if optimizer.lower() == 'rms':
            optimizer = 'rmsprop'
        if optimizer.lower() not in ('adam', 'rmsprop'):
            raise ValueError('unknown optimizer requested!')

        inpDim = dim + (1 if timeDependent else 0)
        if MORvar is not None:
            inpDim += int(np.sum(MORvar.varNum))
        lossOpt = {'integWflag': integPnum != 2}
        lossOpt['isSource'] = not (hasattr(PDE, 'source') and PDE.source == 0.0)   # VarNet.py:184

        self.dim, self.discNum, self.bDiscNum, self.tDiscNum = dim, discNum, bDiscNum, tDiscNum
        self.MORdiscScheme = MORdiscScheme
        self.modelId, self.PDE = modelId, PDE
        self.layerWidth, self.inpDim, self.lossOpt = list(layerWidth), inpDim, lossOpt
        self.activationFun, self.optimizer, self.learning_rate = activationFun, optimizer, learning_rate
        self.processors, self.controller = processors, controller

        self.fixData = FIXData(self, integPnum)
        self.fixData.setInputData(self)
        self.fixData.setFEdata()

        # distributed context: one process per GPU; world_size plays the reference's puNum
        self.rank, self.world, self.dist = 0, 1, None
        self._towers = None
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.dist, self.rank, self.world = dist, dist.get_rank(), dist.get_world_size()
        except ImportError:
            pass
        if isinstance(processors, (list, tuple)) and len(processors) > 1 and self.world == 1:
            # the reference's single-process multi-GPU call (TFModel.py:120-165): this process becomes the
            # controller of one forked child per GPU (varnet_amd/towers.py); it never touches the GPU itself
            from .towers import TowerGroup
            kw = dict(layerWidth=layerWidth, modelId=modelId, activationFun=activationFun, discNum=discNum,
                      bDiscNum=bDiscNum, tDiscNum=tDiscNum, MORdiscScheme=MORdiscScheme, processors=list(processors),
                      controller=controller, integPnum=integPnum, optimizer=optimizer, learning_rate=learning_rate)
            self._towers = TowerGroup(type(self), (PDE,), kw, list(processors))
            self.world = self._towers.world
            self.engine = self.tfData = None
            return
        self._rng = np.random.default_rng(12345)
        self.engine = self._make_engine(processors)
        self.engine.init_params(seed=0)
        # tower gradient SUM (TFModel.py:342-377): by default an RCCL communicator inside the engine, so a
        # step is gradient -> all-reduce -> optimizer on one stream with one host call; VN_COMM=torch keeps the
        # collective in torch.distributed (three host calls per step)
        self.comm, self.comm_why = 'none', ''
        if self.world > 1:
            self.comm = 'torch'
            # auto: in-engine RCCL whenever the ranks own distinct GPUs (nccl backend); rccl: required; try: attempted
            # whatever the backend, falling back to torch.distributed; torch: never
            mode = os.environ.get('VN_COMM', 'auto')
            if hasattr(self.engine, 'comm_init_from_torch') and \
                    (mode in ('rccl', 'try') or (mode == 'auto' and self.dist.get_backend() == 'nccl')):
                # every rank must end up on the SAME route; the bootstrap itself is collective-safe (a rank that cannot
                # load RCCL makes ALL ranks skip it: VNEngine.comm_init_from_torch), so the decision needs no extra vote
                ok, why = self.engine.comm_init_from_torch(self.dist)
                self.comm_why = why                 # why the in-engine communicator was skipped ('' when it is up)
                if ok:
                    self.comm = 'rccl'
                else:
                    if mode == 'rccl':
                        raise RuntimeError('VN_COMM=rccl but the in-engine RCCL communicator did not come up on every rank'
                                           + (': ' + why if why else ''))
                    warnings.warn('in-engine RCCL communicator unavailable (%s): gradient SUM stays in torch.distributed'
                                  % (why or 'another rank failed'))
            else:
                self.comm_why = 'not attempted: VN_COMM=%s on the %s backend' % (mode, self.dist.get_backend())
        fd = self.fixData
        self.engine.set_fe_table(fd.N, fd.dNt, None if fd.integW is None else fd.integW)
        self.tfData = self.engine       # name kept for scripts that poke at `VarNet.tfData`
        from .launch import mark_stage
        mark_stage('engine_ready')      # past the launcher's bootstrap deadline: a rank that trains for hours is healthy

    # -- engine ----------------------------------------------------------------------------
    def _make_engine(self, processors):
        from .engine import VNEngine
        device = 0
        if isinstance(processors, (list, tuple)):
            if len(processors) > 1:
                # deliberate divergence (INTEGRATION.md "Multi-GPU"): TF-1 drives all towers from ONE process
                # (TFModel.py:120-165, 253-289); here every GPU has its own process.  Inside a launched rank the
                # list is accepted and this rank takes its own entry.
                if self.world > 1 and len(processors) == self.world:
                    processors = processors[self.rank]
                else:
                    raise ValueError('processors=%s lists %d GPUs but %d ranks are running: pass one entry per rank'
                                     % (processors, len(processors), self.world))
            else:
                processors = processors[0]
        if isinstance(processors, str):
            kind, _, idx = processors.partition(':')
            if kind.upper() != 'GPU':
                raise ValueError('requested processor %s is unavailable!' % processors)
            device = int(idx or 0)
        elif 'LOCAL_RANK' in os.environ:
            device = int(os.environ['LOCAL_RANK'])
        fd = self.fixData
        return VNEngine(self.dim, self.inpDim, self.layerWidth, self.PDE.timeDependent, fd.integNum,
                        isSource=self.lossOpt['isSource'], integWflag=self.lossOpt['integWflag'],
                        learning_rate=self.learning_rate, device=device, activationFun=self.activationFun,
                        optimizer_name=self.optimizer)

    # -- discretisation ----------------------------------------------------------------------
    def timeDisc(self, tdof=None, rfrac=0, sortflg=True, discTol=None):
        """t nodes linspace(t0+ht, T, tdof), ht=(T-t0)/tdof (VarNet.py:297-338)."""
        PDE = self.PDE
        if not PDE.timeDependent:
            raise Exception('The problem is time-independent!')
        rfrac = min(max(rfrac, 0), 1)
        if tdof is None:
            tdof = self.tDiscNum
        tlim = PDE.tInterval
        ht = (tlim[1] - tlim[0]) / tdof
        tol = ht if discTol is None else float(np.reshape(discTol, -1)[0])
        dof1 = math.floor(tdof * rfrac)
        t1 = np.random.uniform(tlim[0] + tol, tlim[1], dof1)
        t2 = np.linspace(tlim[0] + tol, tlim[1], tdof - dof1)
        t = np.hstack([t1, t2]) if dof1 else t2
        if rfrac > 0 and sortflg:
            t = np.sort(t)
        return ht, np.reshape(t, [tdof, 1])

    def trainingPoints(self, smpScheme='uniform', frac=0.5, addTrainPts=True, suppFactor=1.0):
        """
        Quadrature-point coordinates of every test function (VarNet.py:504-600):
        Input[k*integNum+p, d] = x_k[d] + he[d]*delta[d,p],  Input[., dim] = t_k + ht*delta[-1,p];
        test functions ordered space-major, time-minor.
        """
        if smpScheme == 'optimal':
            return self.optTrainPoints(frac, addTrainPts, suppFactor)
        rfrac = frac if smpScheme == 'random' else 0.
        dim, PDE, fd = self.dim, self.PDE, self.fixData
        domain = PDE.domain
        dof, nt, nT, delta = fd.dof, fd.nt, fd.nT, fd.delta
        if PDE.timeDependent:
            tDiscNum = self.tDiscNum
            ht, t_coord = self.timeDisc(rfrac=rfrac)
        else:
            tDiscNum, t_coord = 1, []
        mesh = domain.getMesh(self.discNum, self.bDiscNum, rfrac=rfrac)
        if smpScheme == 'random' and mesh.dof < dof:
            coord = mesh.coordinates
            while coord.shape[0] < dof:
                coord = uf.vstack([coord, domain.getMesh(self.discNum, self.bDiscNum, rfrac=1.).coordinates])
            mesh.dof, mesh.coordinates = dof, coord[:dof, :]
        he = np.reshape(mesh.he, -1)
        coord = mesh.coordinates
        Coord = np.empty([nT, dim])
        for d in range(dim):
            c = np.repeat(coord[:, d], tDiscNum).reshape(nt, 1) + he[d] * delta[d, :]
            Coord[:, d] = c.reshape(nT)
        if PDE.timeDependent:
            tC = np.tile(t_coord, [dof, 1]) + ht * delta[-1, :]
            Input = np.concatenate([Coord, tC.reshape(nT, 1)], axis=1)
        else:
            Input = Coord
        biInput, biDof = self.biTrainPoints(mesh, t_coord)
        return Input, [], biInput, biDof

    def optTrainPoints(self, frac=0.25, addTrainPts=True, suppFactor=1.0):
        """
        Residual-driven ("optimal") training points (VarNet.py:1696-1868): keep (or thin) the uniform
        test functions and draw the others by rejection sampling with acceptance probability
        |PDE residual| / max|residual on the uniform grid|; optionally shrink the support of the
        added ones by `suppFactor`.  The residual field comes from the device (`vn_residual`).
        """
        dim, PDE, fd = self.dim, self.PDE, self.fixData
        td, domain = PDE.timeDependent, PDE.domain
        feDim, integNum, nt, nT, hVec, delta = fd.feDim, fd.integNum, fd.nt0, fd.nT, fd.hVec, fd.delta
        frac2 = 1 if addTrainPts else (1 - frac) ** (1 / feDim)
        if td:
            tDiscNum2 = math.ceil(frac2 * self.tDiscNum)
            _, t_coord = self.timeDisc(tDiscNum2)
        else:
            tDiscNum2, t_coord = 1, []
        discNum2 = [math.ceil(frac2 * d) for d in self.discNum]
        mesh = domain.getMesh(discNum2, self.bDiscNum)
        input2 = uf.pairMats(mesh.coordinates, t_coord)
        nt1 = math.ceil(frac * nt) if addTrainPts else nt - mesh.dof * tDiscNum2
        scaled = np.abs(suppFactor - 1.0) > 1.e-15
        tole = suppFactor * hVec[:dim] if scaled else None
        tolt = suppFactor * hVec[-1] if (scaled and td) else None

        def resfun(inpuT=None):
            _, resVec, _, _ = self.residual(inpuT)
            return np.abs(resVec)

        def smpfun():
            tc = self.timeDisc(rfrac=1, sortflg=False, discTol=tolt)[1] if td else []
            m = domain.getMesh(self.discNum, self.bDiscNum, rfrac=1, sortflg=False, discTol=tole)
            return uf.pairMats(m.coordinates, tc)

        input1 = uf.rejectionSampling(resfun, smpfun, nt1)
        if addTrainPts:
            nt = nt1 + nt
            nT = nt * integNum
        inpuT = np.vstack([input1, input2])
        coord = inpuT[:, :dim]
        if td and not scaled:                                        # sort by time only for equal supports
            t_coord = inpuT[:, dim:dim + 1]
            ind = np.reshape(np.argsort(t_coord, axis=0), nt)
            coord, t_coord = coord[ind], t_coord[ind]
        elif td:
            t_coord = inpuT[:, dim:dim + 1]
        biInput, biDof, _, _ = self.optBiTrainPoints(frac, addTrainPts)

        he = np.reshape(hVec[:dim], -1)
        suppScale = 1.0
        if scaled:
            suppScale = np.ones([nt, 1])
            suppScale[:nt1, :] = suppFactor
        Coord = np.empty([nT, dim])
        for d in range(dim):
            Coord[:, d] = (np.reshape(coord[:, d], [nt, 1]) + he[d] * delta[d, :] * suppScale).reshape(nT)
        if td:
            tC = t_coord + float(np.reshape(hVec[-1], -1)[0]) * delta[-1, :] * suppScale
            Input = np.concatenate([Coord, tC.reshape(nT, 1)], axis=1)
        else:
            Input = Coord
        if addTrainPts:
            fd.updateOptimData(frac, suppFactor)
        return Input, [], biInput, biDof

    def optBiTrainPoints(self, frac=0.25, addTrainPts=True):
        """Optimal boundary / initial points (VarNet.py:1872-1966): rejection sampling per
        Dirichlet edge and for the initial slice with density (model - label)^2."""
        dim, PDE, fd = self.dim, self.PDE, self.fixData
        td, domain = PDE.timeDependent, PDE.domain
        biDof = fd.biDof0
        frac2 = 1 if addTrainPts else (1 - frac) ** (1 / (fd.feDim - 1))
        t_coord = self.timeDisc(math.ceil(frac2 * self.tDiscNum))[1] if td else []
        discNum2 = [math.ceil(frac2 * d) for d in self.discNum]
        bDisc = self.bDiscNum
        bDisc2 = math.ceil(frac2 * bDisc) if isinstance(bDisc, (int, float)) else bDisc
        mesh = domain.getMesh(discNum2, bDisc2)
        biInput2, biDof2 = self.biTrainPoints(mesh, t_coord)
        if addTrainPts:
            biDof1 = [math.ceil(frac * b) for b in biDof]
        else:
            biDof1 = list(np.array(biDof) - np.array(biDof2))

        def resfun(biInpuT=None):
            if biInpuT is None:
                biInpuT = fd.uniform_biInput
            val = self._model_on(biInpuT)
            return (val - self.biTrainData(biInpuT, biDof)) ** 2

        def smpfun():
            tc = self.timeDisc(rfrac=1, sortflg=False)[1] if td else []
            m = domain.getMesh(self.discNum, self.bDiscNum, rfrac=1, sortflg=False)
            return self.biTrainPoints(m, tc)[0]

        biInput1 = uf.rejectionSampling(resfun, smpfun, biDof1, biDof)
        biDofNew = list(np.array(biDof1) + np.array(biDof2)) if addTrainPts else list(biDof)
        seg1 = uf.listSegment(biInput1, biDof1)
        seg2 = uf.listSegment(biInput2, biDof2)
        out = []
        for i in range(len(biDof1)):
            b = uf.vstack([seg1[i], seg2[i]])
            if td:
                ind = np.reshape(np.argsort(b[:, dim:dim + 1], axis=0), biDofNew[i])
                b = b[ind]
            out.append(b)
        return uf.vstack(out), [int(v) for v in biDofNew], biInput1, biInput2

    def _model_on(self, Input):
        """model(Input) for space-time rows without MOR columns (VarNet.py:1930) -> [n,1]."""
        return self.engine.forward(Input).cpu().numpy().astype(np.float64).reshape(-1, 1)

    def biTrainPoints(self, mesh, t_coord):
        """Dirichlet-edge x time points, then IC points [x,0] (VarNet.py:604-645)."""
        PDE = self.PDE
        bInput, biDof = [], []
        for bInd in range(PDE.domain.bIndNum):
            if PDE.BCtype[bInd] == 'Dirichlet':
                b = uf.pairMats(mesh.bCoordinates[bInd], t_coord)
                biDof.append(len(b))
                bInput.append(b)
        bInput = uf.vstack(bInput)
        iInput = []
        if PDE.timeDependent:
            iInput = np.concatenate([mesh.coordinates, np.zeros([mesh.dof, 1])], axis=1)
            biDof.append(mesh.dof)
        return uf.vstack([bInput, iInput]), biDof

    def biTrainData(self, biInput, biDof, biArg=[]):
        """Labels g/beta on Dirichlet edges, IC(x) on the initial slice (VarNet.py:649-722)."""
        dim, PDE = self.dim, self.PDE
        td = PDE.timeDependent
        nb = PDE.domain.bIndNum
        if uf.isempty(biArg):
            biArg = [{} for _ in range(nb)] + ([{}] if td else [])
        labels, indp, j = [], 0, 0
        ind = 0
        for bInd in range(nb):
            if PDE.BCtype[bInd] != 'Dirichlet':
                continue
            ind = indp + biDof[j]
            j += 1
            beta, g = PDE.BCs[bInd][1], PDE.BCs[bInd][2]
            targ = [biInput[indp:ind, dim][np.newaxis].T] if td else []
            arg = biArg[bInd] if biArg[bInd] is not None else {}
            labels.append(g(biInput[indp:ind, :dim], *targ, **arg) / beta)
            indp = ind
        out = uf.vstack(labels)
        if td:
            arg = biArg[-1] if biArg[-1] is not None else {}
            out = uf.vstack([out, PDE.IC(biInput[ind:, :dim], **arg)])
        return out

    def PDEinpData(self, Input, inpArg=[]):
        """kappa, v, s at the rows of Input (VarNet.py:726-774)."""
        dim, PDE = self.dim, self.PDE
        if uf.isempty(inpArg):
            diffArg, velArg, sourceArg = {}, {}, {}
        else:
            diffArg, velArg, sourceArg = [a if a is not None else {} for a in inpArg]
        targ = [Input[:, dim][np.newaxis].T] if PDE.timeDependent else []
        X = Input[:, 0:dim]
        return (PDE.diffFun(X, *targ, **diffArg), PDE.velFun(X, *targ, **velArg),
                PDE.sourceFun(X, *targ, **sourceArg))

    def MORargExtract(self, batch, MORdiscArg):
        """Keyword arguments of every parametric callable for MOR batch `batch`, and the extra
        network inputs, in the reference's order (VarNet.py:901-1049): BC functions, IC, diff,
        vel, source."""
        PDE = self.PDE
        fi = PDE.MORfunInd
        names = PDE.MORvar.ArgNames
        argInd = self.fixData.MORargInd
        nb = PDE.domain.bIndNum
        inpNN = []

        def kw(find):
            vals = MORdiscArg[find][int(argInd[batch, find]), :]
            inpNN.extend(vals)
            return uf.buildDict(names[find], vals)

        biArg = [{} for _ in range(nb)] + ([{}] if PDE.timeDependent else [])
        if fi['biData']:
            for bInd in range(nb):
                if PDE.BCtype[bInd] == 'Dirichlet' and fi['BCs'][bInd] is not None:
                    biArg[bInd] = kw(fi['BCs'][bInd])
            if fi['IC'] is not None:
                biArg[-1] = kw(fi['IC'])
        inpArg = [{}, {}, {}]
        if fi['inpData']:
            for j, key in enumerate(('diff', 'vel', 'source')):
                if fi[key] is not None:
                    inpArg[j] = kw(fi[key])
        return biArg, inpArg, np.reshape(np.asarray(inpNN, dtype=float), [1, len(inpNN)])

    # -- device-resident data -----------------------------------------------------------------
    def _assemble(self, Input, biInput, biDof, batch, MORdiscArg):
        """Host fp64 assembly of one MOR batch -> dict of device tensors (VarNet.py:778-857)."""
        fd, eng = self.fixData, self.engine
        if MORdiscArg is None:
            biArg, inpArg, MORinp = [], [], None
        else:
            biArg, inpArg, MORinp = self.MORargExtract(batch, MORdiscArg)
        biLabel = self.biTrainData(biInput, biDof, biArg)
        diff, vel, src = self.PDEinpData(Input, inpArg)
        nt, q, dim = fd.nt, fd.integNum, self.dim
        N_rows = dNt_rows = None
        if fd.detJvec:
            # non-uniform supports: per-row tables (VarNetUtility.py:506-523)
            Nr, dNxr, dNtr = fd.rows()
            gcoef = diff * dNxr + vel * Nr                          # VarNet.py:837
            N_rows, dNt_rows = eng.dev(Nr.reshape(-1)), eng.dev(dNtr.reshape(-1))
        else:
            # gcoef = kappa*dNx + v*N  (VarNet.py:837) with the period tables broadcast over test functions
            gcoef = (diff.reshape(nt, q, 1) * fd.dNx[None, :, :] +
                     vel.reshape(nt, q, dim) * fd.N[None, :, None]).reshape(nt * q, dim)
        if MORinp is not None:
            Input = np.hstack([Input, np.tile(MORinp, [Input.shape[0], 1])])
            biInput = np.hstack([biInput, np.tile(MORinp, [biInput.shape[0], 1])])
        return dict(Input=eng.dev(Input), Input_host=Input, gcoef=eng.dev(gcoef),
                    source=eng.dev(src.reshape(-1)) if self.lossOpt['isSource'] else None,
                    biInput=eng.dev(biInput), biLabel=eng.dev(biLabel.reshape(-1)),
                    N_rows=N_rows, dNt_rows=dNt_rows,
                    detJ=eng.dev(np.reshape(fd.detJ, -1)) if fd.detJvec else None)

    def _sync_sampling(self):
        """Towers must draw ONE training set (the reference samples once and slices it per tower,
        VarNetUtility.py:819-857): rank 0 draws a seed from its NumPy stream, every rank re-seeds with it."""
        if self.world == 1 or self.dist is None:
            return
        box = [int(np.random.randint(0, 2 ** 31 - 1)) if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0)
        np.random.seed(box[0])

    def _build_tdata(self, batchNum=None, batchLen=None, smpScheme='uniform', frac=0.5, addTrainPts=True,
                     suppFactor=1.0):
        fd = self.fixData
        if smpScheme != 'uniform':
            self._sync_sampling()
        Input, _, biInput, biDof = self.trainingPoints(smpScheme, frac, addTrainPts, suppFactor)
        if smpScheme == 'optimal' and not addTrainPts:
            fd.biDof = [int(b) for b in biDof]
        MORdiscArg = fd.MORdiscArg
        mor = [self._assemble(Input, biInput, biDof, b, MORdiscArg) for b in range(fd.MORbatchNum)]
        return ManageTrainData(self, mor, batchNum, batchLen)

    # -- loss components / weights ----------------------------------------------------------------
    def _allreduce(self, t):
        if self.dist is not None and self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)

    def _gather_lossVec(self, lv, tData, bi):
        """Loss field of mini-batch `bi` as the reference's controller sees it: the towers' fields concatenated in
        tower order (`tf.concat([compTowers[pu].lossVec ...], axis=0)`, TFModel.py:319), one entry per test function
        of the blocks `ManageTrainData.block(bi, pu)`.  One all-gather of `batchLen` floats per rank (ragged and empty
        last blocks are padded for the collective and cut afterwards); every rank returns the same column."""
        torch = self.engine.torch
        if self.world == 1 or self.dist is None:
            return lv.cpu().numpy().reshape(-1, 1)
        L = int(tData.batchLen)
        # gloo gathers host tensors only; RCCL gathers where the field lies
        dev = lv.device if self.dist.get_backend() == 'nccl' else torch.device('cpu')
        mine = torch.zeros(L, dtype=lv.dtype, device=dev)
        mine[:lv.numel()] = lv.reshape(-1).to(dev)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        cols = []
        for pu in range(self.world):
            n0, n1 = tData.block(bi, pu)
            cols.append(parts[pu][:n1 - n0])
        return torch.cat(cols).cpu().numpy().reshape(-1, 1)

    def splitLoss(self, tData, W=None):
        """BC, IC and variational loss summed over MOR batches (VarNet.py:1053-1090)."""
        if W is None:
            W = np.eye(3)
        eng, fd = self.engine, self.fixData
        torch = eng.torch
        comp = np.zeros([3, 1])
        lossVec = [] if fd.lossVecflag else None
        for mb in range(fd.MORbatchNum):
            tData.select_mor(mb)
            var = 0.0
            lv_b = []
            for bi in range(tData.batchNum):
                out, lv = eng.eval_loss(tData.engine_batch(mb, bi), lossVec=fd.lossVecflag)
                bc, ic = out[1], out[2]
                var += out[3]
                if lv is not None:
                    lv_b.append(self._gather_lossVec(lv, tData, bi))
            if self.world > 1:
                t = torch.tensor([var], dtype=torch.float64, device=eng.device)
                self._allreduce(t)
                var = float(t.item())
            comp += np.array([[bc, ic, var]]).T
            if lossVec is not None and lv_b:
                lossVec.append(np.vstack(lv_b))
        return np.matmul(W, comp), tData, lossVec

    def trainWeight(self, weight, tData, normalizeW=False, useOriginalW=False, lossTot=1.e6):
        """Initial penalty weights (VarNet.py:1094-1146): default branch scales `weight` so the
        weighted initial loss is 1e6."""
        td = self.PDE.timeDependent
        lossVal, tData, _ = self.splitLoss(tData)
        lossVal = np.reshape(lossVal, 3)
        lossTmp = lossVal if td else np.array([lossVal[0], lossVal[2]])
        if useOriginalW:
            trainW = np.array(weight, dtype=float)
        elif normalizeW:
            nw = len(weight)
            W = np.tile(weight, [nw, 1]) / np.reshape(weight, [nw, 1])
            W = np.sum(W, axis=1, keepdims=True) * np.reshape(lossTmp, [nw, 1])
            trainW = np.reshape(lossTot / W, nw)
        else:
            trainW = lossTot / np.sum(np.array(weight) * np.array(lossTmp)) * np.array(weight, dtype=float)
        if not td:
            trainW = np.array([trainW[0], 0., trainW[1]])
        s = 'Training weight information:\n'
        s += '\tboundary condition loss value: %.4e\n\tinitial condition loss value: %.4e\n' % (lossVal[0], lossVal[1])
        s += '\tintegral loss value: %.4e\n\trequested weight on each term: %s\n' % (lossVal[2], str(weight))
        s += '\tcorresponding training weights: %s\n\n' % np.array2string(np.array(trainW), precision=4)
        if self.trainRes.verbose and self.rank == 0:
            print(s)
        self.trainRes.writeCase(s)
        return np.array(trainW, dtype=float), tData, lossVal

    # -- training -------------------------------------------------------------------------------------
    def optimIter(self, tData, mb, loss_acc):
        """One pass over the mini-batches of MOR batch mb (VarNetUtility.py:1021-1047):
        gradient, tower SUM (all-reduce), TF-1 Adam; the pre-update loss is added to the
        device scalar `loss_acc`."""
        eng = self.engine
        P = eng.P
        if (self.world == 1 or self.comm == 'rccl') and hasattr(eng, 'train_epoch'):
            # the whole pass in one host call (no per-mini-batch Python / launch-queue gaps); with towers the
            # engine's own RCCL communicator sums the gradient between the two kernels
            ids = tData._epoch_ids.get(mb) if hasattr(tData, '_epoch_ids') else None
            if ids is None:
                if not hasattr(tData, '_epoch_ids'):
                    tData._epoch_ids = {}
                ids = tData._epoch_ids[mb] = tuple(tData.engine_batch(mb, bi) for bi in range(tData.batchNum))
            eng.train_epoch(ids, loss_acc)
            return
        gb = eng.bind_grad_buffer()
        for bi in range(tData.batchNum):
            eng.grad(tData.engine_batch(mb, bi))
            self._allreduce(gb)
            eng.apply()
            loss_acc += gb[P]

    def _choose_formulation(self, tData, dedup, shuffleData=False):
        """Apply `train(dedup=...)` to a training set; returns {'requested', 'on', 'unique_points', 'reason'} and records the
        decision in caseData.txt.  A request that cannot be served is a warning (`dedup=True`) or a recorded sentence
        ('auto'), never a silent row-wise run."""
        state = {'requested': dedup, 'on': False, 'unique_points': 0, 'reason': None}
        if dedup is False:
            state['reason'] = 'dedup=False was requested'
        elif shuffleData:
            state['reason'] = 'shuffleData=True: ' + \
                'the mini-batches are shuffled (the point map of every block would have to be rebuilt at each shuffle)'
        else:
            state['reason'] = tData.dedup_applies()
            if state['reason'] is None and dedup == 'auto' and not tData.dedup_pays():
                state['reason'] = ('the step is too small for it to pay (three more kernel launches and a second prologue '
                                   'against the work saved: DEDUP_MODEL); dedup=True forces it')
            if state['reason'] is None:
                state['unique_points'] = int(tData.enable_dedup())
                state['on'] = state['unique_points'] > 0
                state['reason'] = tData.dedup_reason
        if state['on']:
            msg = ('interior term: de-duplicated formulation, %d unique quadrature points for %d rows per epoch '
                   '(dedup=%r)\n\n' % (state['unique_points'], self.fixData.nT * self.fixData.MORbatchNum, dedup))
        else:
            msg = 'interior term: row-wise formulation (dedup=%r): %s\n\n' % (dedup, state['reason'])
            if dedup is True:
                warnings.warn('dedup=True cannot be served, training row-wise: ' + state['reason'])
        if getattr(self, 'trainRes', None) is not None:
            if self.trainRes.verbose and self.rank == 0:
                print(msg)
            self.trainRes.writeCase(msg)
        return state

    def train(self, folderpath, weight=None, smpScheme='uniform', epochNum=500000, tol=1.e-1,
              verbose=True, saveFreq=100, pltReplace=True, saveMORdata=False, frac=None,
              addTrainPts=True, suppFactor=1.0, multiTrainUpd=False, trainUpdelay=2e4, tolUpd=0.01,
              reinitrain=True, updateWeights=False, normalizeW=False, adjustWeight=False,
              useOriginalW=False, batchNum=None, batchLen=None, shuffleData=False, shuffleFreq=1,
              dedup='auto', lossLag=None):
        """Training loop of /root/reference/VarNet.py:1197-1421 (uniform, random and residual-driven
        "optimal" sampling with re-initialisation and re-weighting)."""
        if self._towers is not None:                  # controller of forked towers: every tower runs the loop
            kw = {k: v for k, v in locals().items() if k not in ('self', 'folderpath')}
            self.folderpath = folderpath
            self.trainRes = self._towers.call('train', folderpath, **kw)
            return self.trainRes
        if uf.isnone(folderpath) or uf.isempty(folderpath):
            raise ValueError('a folder path must be provided to backup the trained model!')
        self.folderpath = folderpath
        td = self.PDE.timeDependent
        if weight is None:
            weight = [1., 1., 1.] if td else [1., 1.]
        elif td and len(weight) != 3:
            raise ValueError('weight dimension does not match!')
        elif not td and len(weight) != 2:
            raise ValueError('weight dimension does not match!')
        if smpScheme not in ('uniform', 'random', 'optimal'):
            raise ValueError('sampling scheme is not valid!')
        if updateWeights:
            raise NotImplementedError('updateWeights=True is broken in the reference (VarNet.py:1373)')
        self.smpScheme = smpScheme
        if frac is None:
            frac = 0.50 if addTrainPts else 0.25
        if batchNum is None and batchLen is None and shuffleData:
            warnings.warn('shuffling data is possible for batch-optimization, setting \'shuffleData\' to False!')
            shuffleData = False
        argDict = {k: v for k, v in locals().items() if k != 'self'}

        if not addTrainPts and np.abs(suppFactor - 1.0) > 1.e-15:
            warnings.warn('\'suppFactor\' is set to 1.0 since the number of training points does not change!')
            suppFactor = 1.0
        eng, fd = self.engine, self.fixData
        torch = eng.torch
        tData = self._build_tdata(batchNum, batchLen)        # first set is always uniform (VarNet.py:1300)
        trainRes = TrainResult(folderpath if self.rank == 0 else None, fd.cEx is not None, verbose, saveFreq, pltReplace)
        trainRes.initializeCase(self, argDict)
        trainRes._plot_last = time.perf_counter()            # the convergence plots are refreshed at most every `plotEvery` seconds and
        self.trainRes = trainRes                             # flushed when train() returns: a short run draws them once (1.1 s at 300 dpi)
        # Formulation of the interior term (extension; the arithmetic is the reference's either way, VarNet.py:576-588 and
        # TFModel.py:653-664): 'auto' (default) evaluates the network once per UNIQUE quadrature point wherever that applies
        # and pays, True asks for it and warns when it cannot be had, False keeps one evaluation per (test function, point) row.
        if dedup not in (True, False, 'auto'):
            raise ValueError('dedup must be True, False or \'auto\'!')
        self.dedup_state = self._choose_formulation(tData, dedup, shuffleData)

        def set_train_weights(tD, wts):
            eng.set_weights([1.0, 1.0, 1.0])
            tW, tD, lv = self.trainWeight(wts, tD, normalizeW, useOriginalW)
            w_e = tD.towerWeights(tW)                                 # VarNetUtility.py:900-901
            eng.set_weights(w_e)
            return tW, w_e, lv

        trainW, w_eff, lossVal = set_train_weights(tData, weight)
        # what the reference stores is the array `updateDictFields('trainW', ...)` has divided IN PLACE by batchNum * puNum
        # (VarNet.py:1324-1325 keeps a reference to it, VarNetUtility.py:900-901 rewrites its BC/IC entries): the record in
        # trainData.vn therefore holds the per-feed weights, not trainWeight's return value
        trainRes.trainWeight = np.array(w_eff, dtype=float)
        trainRes.lossComp.append(lossVal)
        self.tData = tData
        tData0 = tData                                         # uniform set for universal cost comparison

        min_loss = float('inf')
        epoch_time = 0.0
        tp_epoch, tp_updates = 1, 0
        resVal = err = lossComp = lossVec = None
        # `lossLag` (extension): the reference reads the loss back after every epoch (VarNetUtility.py:1044); with lossLag = k > 1
        # up to k epochs are enqueued before ONE read-back of their k losses, so small problems (an epoch of 30 us of GPU work
        # behind a 20 us host round trip) keep the GPU busy.  It is a read-back SCHEDULE, not another algorithm: blocks end at
        # every epoch that acts on the state (saveFreq monitors / checkpoints, shuffles), and since round 6 the stopping test
        # `loss < tol` is exact too -- the engine snapshots its state on the device at the start of a block, and when the test
        # fires inside the block the state is rolled back and the epochs up to the one that met the tolerance are replayed (the
        # steps are bitwise reproducible): same losses, same checkpoints, same final parameters and step count as k = 0
        # (tests/test_varnet_host.py::test_loss_lag_is_only_a_readback_schedule).  Default (None): 8 with an engine that can
        # snapshot, else 0.
        can_snap = hasattr(eng, 'state_snapshot')
        if lossLag is None:
            lossLag = 8 if can_snap else 0
        # Non-uniform sampling: the re-draw test of VarNet.py:1385-1421 runs after every epoch, but what it looks at -- the losses
        # sampled at the monitors, the epochs since the last re-draw -- changes only at monitor epochs (where blocks end anyway) and
        # with the epoch count, so the first epoch at which it can fire inside a block is known when the block is sized
        # (`redraw_at` below) and the block ends there.  A schedule without the engine's snapshot cannot make the stop exact: 0.
        lag = max(0, int(lossLag)) if (smpScheme == 'uniform' or can_snap) else 0

        def redraw_at(e0):
            """First epoch >= e0 at which the re-draw test fires if no monitor intervenes, or None."""
            if smpScheme == 'uniform' or not (multiTrainUpd or tp_updates == 0):
                return None
            t_loss = np.array(trainRes.loss[-5:])
            if len(t_loss) == 0:
                return None
            conv = t_loss[:-1] - t_loss[1:]
            if not (np.sum(conv[conv > 0]) / t_loss[-1] < tolUpd):
                return None
            return max(e0, int(math.ceil(tp_epoch + trainUpdelay - 1)))
        loss_buf = torch.zeros(max(lag, 1), dtype=torch.float32, device=eng.device)
        epoch = 1
        done = False
        while epoch <= epochNum and not done:
            nblk = 1
            if lag > 1:
                nblk = min(lag, epochNum - epoch + 1, saveFreq - (epoch - 1) % saveFreq)
                if shuffleData:
                    nblk = min(nblk, shuffleFreq - (epoch - 1) % shuffleFreq)
                e_f = redraw_at(epoch)
                if e_f is not None:
                    nblk = max(1, min(nblk, e_f - epoch + 1))
            t0 = time.perf_counter()
            loss_buf.zero_()
            if nblk > 1 and can_snap:
                eng.state_snapshot()                                 # device-side, on the engine's stream, no synchronisation
            for i in range(nblk):
                for mb in range(fd.MORbatchNum):
                    tData.select_mor(mb)
                    self.optimIter(tData, mb, loss_buf[i])
            # The reshuffle that follows this epoch (VarNetUtility.py:957-1017 at VarNet.py:1354) is prepared BEFORE the
            # loss is read back: its host work and gathers then run while the GPU is still busy with the epoch.  Safe:
            # engine and gathers share one stream (old buffers are reused in stream order), the NumPy draws keep their
            # order (no other consumer between the two points with uniform sampling).
            early_shuffle = (nblk == 1 and smpScheme == 'uniform' and shuffleData and epoch % shuffleFreq == 0)
            if early_shuffle:
                tData.shuffleTrainData()
            losses = loss_buf[:nblk].tolist()                        # one host sync per block (per epoch when lossLag = 0)
            if nblk > 1 and can_snap:
                hit = next((i for i in range(nblk - 1) if float(losses[i]) < tol), None)
                if hit is not None:
                    # the stopping test fires at epoch first + hit, inside the block: back to the block's start, then exactly the
                    # epochs the reference would have run (their losses are the ones already read)
                    eng.state_rollback()
                    scratch = torch.zeros(1, dtype=torch.float32, device=eng.device)
                    for i in range(hit + 1):
                        for mb in range(fd.MORbatchNum):
                            tData.select_mor(mb)
                            self.optimIter(tData, mb, scratch[0])
            blk_time = time.perf_counter() - t0
            first = epoch
            for i in range(nblk):
                epoch = first + i
                current_loss = float(losses[i])
                epoch_time += blk_time / nblk
                if shuffleData and epoch % shuffleFreq == 0 and not early_shuffle:
                    tData.shuffleTrainData()

                if epoch % saveFreq == 0:
                    if min_loss > current_loss:
                        min_loss = current_loss
                        self.saveModel(epoch)
                    resVal, _, err, _ = self.residual()
                    eng.set_weights([1.0, 1.0, 1.0])
                    if tData0 is not tData:
                        tData0.activate()
                    lossComp, _, lossVec = self.splitLoss(tData0)       # VarNet.py:1365
                    if tData0 is not tData:
                        tData.activate()
                    eng.set_weights(w_eff)
                trainRes.iterOutput(epoch, current_loss, min_loss, epoch_time, resVal, err, lossComp, lossVec)

                if current_loss < tol:
                    trainRes.writeCase('Training completed!')
                    if verbose and self.rank == 0:
                        print('Training completed!')
                    done = True
                    break

                # regenerate the training set (VarNet.py:1385-1421)
                if smpScheme != 'uniform' and (multiTrainUpd or tp_updates == 0) and (epoch - tp_epoch) >= (trainUpdelay - 1):
                    t_loss = np.array(trainRes.loss[-5:])                # sampled every saveFreq epochs, as the reference
                    tp_conv = t_loss[:-1] - t_loss[1:]
                    tp_conv = np.sum(tp_conv[tp_conv > 0])
                    if len(t_loss) > 0 and tp_conv / t_loss[-1] < tolUpd:
                        min_loss = float('inf')
                        tp_epoch = epoch
                        tp_updates += 1
                        trainRes.inpIter.append(epoch)
                        tData = self._build_tdata(batchNum, batchLen, smpScheme, frac, addTrainPts, suppFactor)
                        self.tData = tData
                        msg = '\n\n==========================================================\nTraining points updated.\n\n'
                        if reinitrain:
                            eng.init_params(seed=tp_updates)            # global_variables_initializer, VarNet.py:1412
                            msg += 'trainable variables reinitialized.\n\n'
                        if verbose and self.rank == 0:
                            print(msg)
                        trainRes.writeCase(msg)
                        if adjustWeight:
                            weight = [5 * wv for wv in weight[:-1]] + [weight[-1]]
                        trainW, w_eff, _ = set_train_weights(tData, weight)
            epoch = first + nblk
        trainRes.flushPlots()
        return trainRes

    # -- checkpoints ----------------------------------------------------------------------------------
    # Checkpoint layout.  tf.train.Saver writes every global variable of the graph under its TF name
    # (TFModel.py:307; VarNet.py:1362): the Keras kernels / biases `dense_<i>/kernel` [in,out], `dense_<i>/bias`,
    # `output/kernel`, `output/bias` (TFModel.py:208-242), the Adam slots `<var>/Adam` (m) and `<var>/Adam_1` (v),
    # `beta1_power`, `beta2_power` and the step counter.  `best_model-<epoch>.npz` holds exactly those arrays under
    # those names (RMSProp: `<var>/RMSProp` = mean square, `<var>/RMSProp_1` = momentum), so a TF-side converter is a
    # loop over names.  TF's own files (`best_model-<n>.index/.meta/.data-*`) cannot be restored here -- reading them
    # needs TensorFlow -- and loadModel says so instead of reporting "nothing found" (Checkpoints).
    def _var_names(self):
        names = ['dense_%d' % i for i in range(len(self.layerWidth))] + ['output']
        return names

    def checkpoint_arrays(self):
        """{TF variable name: array} for the current engine state (see the layout note above)."""
        buf = np.asarray(self.engine.export_state(), dtype=np.uint8)
        step = int(buf[:8].view(np.int64)[0])
        P = self.engine.P
        flat = buf[8:].view(np.float32)
        theta, m, v = flat[:P], flat[P:2 * P], flat[2 * P:3 * P]
        slot = ('RMSProp_1', 'RMSProp') if str(self.optimizer).lower() == 'rmsprop' else ('Adam', 'Adam_1')
        out, off, fan = {}, 0, self.inpDim
        for name, h in zip(self._var_names(), self.layerWidth + [1]):
            for part, n, shp in (('kernel', fan * h, (fan, h)), ('bias', h, (h,))):
                key = '%s/%s' % (name, part)
                out[key] = theta[off:off + n].reshape(shp).copy()
                out[key + '/' + slot[0]] = m[off:off + n].reshape(shp).copy()
                out[key + '/' + slot[1]] = v[off:off + n].reshape(shp).copy()
                off += n
            fan = h
        out['global_step'] = np.int64(step)
        if slot[0] == 'Adam':
            out['beta1_power'] = np.float32(0.9 ** (step + 1))       # TF keeps beta^(t+1) for the next update
            out['beta2_power'] = np.float32(0.999 ** (step + 1))
        return out

    def restore_arrays(self, arrays):
        """Inverse of `checkpoint_arrays`: load {TF variable name: array} into the engine."""
        slot = ('RMSProp_1', 'RMSProp') if str(self.optimizer).lower() == 'rmsprop' else ('Adam', 'Adam_1')
        th, m, v, fan = [], [], [], self.inpDim
        for name, h in zip(self._var_names(), self.layerWidth + [1]):
            for part, shp in (('kernel', (fan, h)), ('bias', (h,))):
                key = '%s/%s' % (name, part)
                if key not in arrays:
                    raise ValueError('checkpoint does not match the network architecture (no variable %s)!' % key)
                a = np.asarray(arrays[key], dtype=np.float32)
                if a.shape != shp:
                    raise ValueError('checkpoint does not match the network architecture (%s is %s, expected %s)!'
                                     % (key, a.shape, shp))
                th.append(a.reshape(-1))
                m.append(np.asarray(arrays.get(key + '/' + slot[0], np.zeros(shp)), dtype=np.float32).reshape(-1))
                dflt = np.ones(shp) if slot[1] == 'RMSProp' else np.zeros(shp)
                v.append(np.asarray(arrays.get(key + '/' + slot[1], dflt), dtype=np.float32).reshape(-1))
            fan = h
        step = np.array([int(arrays['global_step']) if 'global_step' in arrays else 0], dtype=np.int64)
        blob = np.concatenate([step.view(np.uint8), np.concatenate(th + m + v).astype(np.float32).view(np.uint8)])
        self.engine.import_state(blob)

    def saveModel(self, epoch):
        """`saver.save(sess, 'best_model', global_step=epoch)` with max_to_keep=2 (TFModel.py:307,
        VarNet.py:1359-1362) -> `best_model-<epoch>.npz` + the `checkpoint` text file TF keeps beside it."""
        if self.rank != 0:
            return
        path = os.path.join(self.folderpath, 'best_model-%d.npz' % epoch)
        np.savez(path, **self.checkpoint_arrays())
        kept = getattr(self, '_ckpts', [])
        kept.append(path)
        while len(kept) > 2:
            old = kept.pop(0)
            if os.path.exists(old):
                os.remove(old)
        self._ckpts = kept
        self._write_checkpoint_file(self.folderpath, epoch, [int(os.path.basename(k)[len('best_model-'):-4]) for k in kept])

    @staticmethod
    def _write_checkpoint_file(folderpath, current, kept):
        with open(os.path.join(folderpath, 'checkpoint'), 'w') as f:
            f.write('model_checkpoint_path: ' + repr(os.path.join(folderpath, 'best_model-%d' % current)))
            for n in kept:
                f.write('\nall_model_checkpoint_paths: ' + repr(os.path.join(folderpath, 'best_model-%d' % n)))
            f.write('\n')

    def loadModel(self, iterNum=None, folderpath=None):
        """Restore the newest (or requested) `best_model-<n>` checkpoint (VarNet.py:1426-1506): without `iterNum`
        the stored iterations are tried newest first; the `checkpoint` file is rewritten to point at the restored
        one, as the reference does."""
        if self._towers is not None:
            if folderpath is None and hasattr(self, 'folderpath'):
                folderpath = self.folderpath
            return self._towers.call('loadModel', iterNum, folderpath)
        if folderpath is None:
            if not hasattr(self, 'folderpath'):
                raise ValueError('\'folderpath\' must be provided!')
            folderpath = self.folderpath
        else:
            self.folderpath = folderpath
            self.trainRes = TrainResult(folderpath)
            if os.path.exists(os.path.join(folderpath, 'trainData.vn')):
                self.trainRes.loadData()
        tf_only = []
        if iterNum is None:
            nums = []
            for f in os.listdir(folderpath):
                if f.startswith('best_model-') and f.endswith('.npz'):
                    nums.append(int(f[len('best_model-'):-4]))
                elif f.startswith('best_model-') and f.endswith('.index'):
                    tf_only.append(f)
            nums.sort(reverse=True)
        else:
            nums = [iterNum]
            if os.path.isfile(os.path.join(folderpath, 'best_model-%d.index' % iterNum)):
                tf_only.append('best_model-%d.index' % iterNum)
        for n in nums:
            path = os.path.join(folderpath, 'best_model-%d.npz' % n)
            if os.path.isfile(path):
                z = np.load(path)
                if 'state' in z.files:                               # files written by round-1 builds
                    if list(z['layerWidth']) != self.layerWidth or int(z['inpDim']) != self.inpDim:
                        raise ValueError('checkpoint does not match the network architecture!')
                    self.engine.import_state(z['state'])
                else:
                    self.restore_arrays({k: z[k] for k in z.files})
                if self.rank == 0:
                    self._write_checkpoint_file(folderpath, n, [n])
                return n
        if tf_only:
            raise ValueError('only TensorFlow saver files (%s ...) were found: they cannot be read without TensorFlow; '
                             'convert them to best_model-<n>.npz with the variable names of '
                             'VarNet.checkpoint_arrays() (INTEGRATION.md, "Checkpoints")' % tf_only[0])
        raise ValueError('no restorable checkpoint data found!')

    def saveNNparam(self, dpOut=False, matOut=False, verbose=False, timeFirst=False, path=None):
        """
        Trained kernels and biases per layer (VarNet.py:2179-2260): list of `[W, b]` with `W [out,in]`,
        `b [out,1]` so that `o = W i + b`; `timeFirst` moves the temporal input column in front of the
        spatial ones in the first layer.  `matOut` writes `NN_parameters/W<n>.mat, B<n>.mat` (MATLAB),
        `dpOut` the Diffpack-readable `W<n>.m, B<n>.m`; `path` additionally writes one `.npz`.
        """
        if self._towers is not None:
            return self._towers.call('saveNNparam', dpOut, matOut, verbose, timeFirst, path)
        flat = self.engine.get_params()
        td = self.PDE.timeDependent
        if not td:
            timeFirst = False
        folder = None
        if dpOut or matOut:
            folder = os.path.join(self.trainRes.folderpath, 'NN_parameters')
            os.makedirs(folder, exist_ok=True)
        layers, npz, off, fan = [], {}, 0, self.inpDim
        for l, h in enumerate(self.layerWidth + [1]):
            W = flat[off:off + fan * h].reshape(fan, h).T.copy()
            off += fan * h
            b = flat[off:off + h].reshape(h, 1).copy()
            off += h
            fan = h
            if l == 0 and timeFirst:
                dim = self.dim
                Wt = W.copy()
                W[:, 0] = Wt[:, dim]
                W[:, 1:dim + 1] = Wt[:, :dim]
            layers.append([W, b])
            npz['W%d' % l], npz['b%d' % l] = W.T.copy(), b[:, 0].copy()
            if verbose:
                print('Layer %d: weight %s, bias %s' % (l, W.shape, b.shape))
            if dpOut:
                uf.mat2diffpack(os.path.join(folder, 'W%d.m' % (l + 1)), 'W%d' % (l + 1), W)
                uf.mat2diffpack(os.path.join(folder, 'B%d.m' % (l + 1)), 'B%d' % (l + 1), b)
            if matOut:
                import scipy.io as spio
                spio.savemat(os.path.join(folder, 'W%d.mat' % (l + 1)), {'W%d' % (l + 1): W})
                spio.savemat(os.path.join(folder, 'B%d.mat' % (l + 1)), {'B%d' % (l + 1): b})
        if path is not None:
            np.savez(path, **npz)
        return layers

    # -- reporting --------------------------------------------------------------------------------------
    def simRes(self, batch=None, tcoord=None, plotpath=None, pltFrmt='png', plot=False):
        """
        Simulation results on the plotting grid (/root/reference/VarNet.py:1970-2175): for every time in `tcoord`
        (default 5 snapshots over the time interval) the fields the reference draws -- approximate solution `cApp`,
        exact solution `cEx` and error `cErr` (when the PDE has `cEx`), strong residual `res`, interpolated loss field
        `lossField` (when the last training run kept `lossVec`) -- as arrays: [51, 51] contour fields in 2D, curves on
        the 51-point x grid in 1D.  Returned as {'t': [...], 'grid': ContourPlot, 'cApp': [...], ..., 'l2Err': [...]}.
        With `plot=True` (and matplotlib present) the same figures are also written under `plotpath` with the
        reference's file names (cApp-t=0.00s.png, cEx-..., cErr-..., res-..., lossField-... / cApp.png, cErr.png,
        residual.png, lossField.png in 1D).
        """
        from .contour import ContourPlot
        if self._towers is not None:
            out = self._towers.call('simRes', batch, tcoord, plotpath, pltFrmt, plot)
            out['grid'] = ContourPlot(self.PDE.domain, self.PDE.tInterval if self.PDE.timeDependent else None)
            return out
        if not hasattr(self, 'trainRes') and uf.isnone(plotpath) and plot:
            raise ValueError('\'plotpath\' must be provided!')
        elif uf.isnone(plotpath) and hasattr(self, 'trainRes'):
            plotpath = self.trainRes.plotpath
        if pltFrmt not in ('png', 'jpg', 'pdf', 'eps'):
            raise ValueError('invalid plot format!')
        suffix = ('.' if uf.isnone(batch) else '-b=' + str(int(batch)) + '.') + pltFrmt
        dim, PDE = self.dim, self.PDE
        td, tInterval, cExact = PDE.timeDependent, PDE.tInterval, PDE.cEx
        if td and uf.isnone(tcoord):
            tcoord = np.linspace(tInterval[0], tInterval[1], num=5)
        elif not td:
            tcoord = [0.]
        if dim > 2:
            raise ValueError('simRes is available for 1D and 2D domains!')
        cp = ContourPlot(PDE.domain, tInterval if td else None)
        cAppFun = lambda x, t=None: self.evaluate(x, t, batch)

        def resFun(x, t=None):
            Input = np.concatenate([x, t * np.ones([len(x), 1])], axis=1) if td else x
            return self.residual(Input, None, batch)[1]

        cExFun = cErrFun = None
        if not uf.isnone(cExact):
            cExFun = (lambda x, t: cExact(x, t * np.ones([len(x), 1]))) if td else (lambda x, t=None: cExact(x))
            cErrFun = lambda x, t=None: cExFun(x, t) - cAppFun(x, t)
        lossFun = None
        lv = getattr(getattr(self, 'trainRes', None), 'lossVec', None)
        if not uf.isnone(lv) and len(lv) > 0:
            from scipy import interpolate
            lossVec = lv[0] if uf.isnone(batch) else lv[batch]
            ui = self.fixData.uniform_input
            if ui.shape[1] == 1:
                lossField = lambda X: np.interp(X[:, 0], ui[:, 0], np.reshape(lossVec, -1), left=0.0, right=0.0).reshape(-1, 1)
            else:
                lossField = interpolate.LinearNDInterpolator(ui, lossVec, fill_value=0.0)
            lossFun = lambda x, t=None: lossField(uf.hstack([x, t * np.ones([len(x), 1])]) if td else x)
        funs = {'cApp': cAppFun, 'cEx': cExFun, 'cErr': cErrFun, 'res': resFun, 'lossField': lossFun}
        out = {'t': [float(t) for t in tcoord], 'grid': cp, 'l2Err': []}
        for name, f in funs.items():
            if f is None:
                continue
            if dim == 1:
                out[name] = [np.asarray(cp.snap(f, float(t))[1], dtype=float).reshape(-1, 1) for t in tcoord]
            else:
                out[name] = [cp.field(f, float(t) if td else None) for t in tcoord]
        if cExFun is not None:
            out['l2Err'] = [uf.l2Err(e, a) for e, a in zip(out['cEx'], out['cApp'])]
        if plot:
            self._simres_figures(out, cp, plotpath, suffix)
        return out

    def _simres_figures(self, out, cp, plotpath, suffix):
        """Draw and save what simRes computed, with the reference's file names (VarNet.py:2066-2172)."""
        try:
            import matplotlib
            matplotlib.use('Agg')
            import matplotlib.pyplot as plt
        except ImportError:
            warnings.warn('matplotlib is not available: simRes figures are not written')
            return
        os.makedirs(plotpath, exist_ok=True)
        ts = out['t']
        if self.dim == 1:
            names = {'cErr': 'cErr', 'res': 'residual', 'lossField': 'lossField'}
            if 'cEx' in out:
                for i, t in enumerate(ts):
                    plt.figure()
                    plt.plot(cp.x_coord, out['cEx'][i], 'b')
                    plt.plot(cp.x_coord, out['cApp'][i], 'r')
                    plt.legend(['exact solution', 'approximate solution'])
                    plt.savefig(os.path.join(plotpath, 'cApp-' + 't={0:.2f}s'.format(t) + suffix), dpi=300)
                    plt.close()
            else:
                names['cApp'] = 'cApp'
            for key, fname in names.items():
                if key not in out:
                    continue
                plt.figure()
                for i, t in enumerate(ts):
                    plt.plot(cp.x_coord, out[key][i])
                plt.legend(['t={0:.2f}s'.format(t) for t in ts])
                plt.savefig(os.path.join(plotpath, fname + suffix), dpi=300)
                plt.close()
        else:
            for key in ('cApp', 'cEx', 'cErr', 'res', 'lossField'):
                if key not in out:
                    continue
                for i, t in enumerate(ts):
                    plt.figure()
                    c = plt.contourf(cp.xx, cp.yy, out[key][i])
                    plt.colorbar(c)
                    plt.axis('scaled')
                    plt.savefig(os.path.join(plotpath, key + '-' + 't={0:.2f}s'.format(t) + suffix), dpi=300)
                    plt.close()

    # -- evaluation -------------------------------------------------------------------------------------
    def _mor_columns(self, batch, n):
        fd = self.fixData
        if self.PDE.MORvar is None:
            return None, [], []
        biArg, inpArg, MORinp = self.MORargExtract(batch, fd.MORdiscArg)
        return np.tile(MORinp, [n, 1]), biArg, inpArg

    def evaluate(self, x=None, t=None, batch=None, MORarg=None):
        """NN approximation of the PDE solution at (x,t[,mu]) (VarNet.py:1510-1595) -> [n,1]."""
        if self._towers is not None:
            return self._towers.call('evaluate', x, t, batch, MORarg)
        dim, PDE, fd = self.dim, self.PDE, self.fixData
        td = PDE.timeDependent
        MORvar = PDE.MORvar
        if x is None:
            if (td and t is None) or not td:
                Input = fd.uniform_input
                dof = Input.shape[0]
            else:
                dof = fd.dof
                x = fd.uniform_input[:dof, :dim]                     # as the reference (VarNet.py:1545)
        elif shape(x)[1] != dim:
            raise ValueError('spatial coordinates dimension does not match domain!')
        else:
            dof = shape(x)[0]
        if td and t is not None and not (size(t) == 1 or shape(t)[0] == dof):
            raise ValueError('temporal discretrization does not match spatial discretization!')
        if td and t is not None and size(t) == 1:
            t = float(np.reshape(t, -1)[0]) * np.ones([dof, 1])
        if MORvar is not None and batch is None and MORarg is None:
            raise ValueError('batch number or argument values must be given for MOR!')
        if MORvar is not None and batch is None and shape(MORarg)[1] != self.inpDim - fd.feDim:
            raise ValueError('MOR argument dimension does not match the NN input size!')
        if MORvar is not None and batch is not None and batch > fd.MORbatchNum - 1:
            raise ValueError('requested batch number is higher than total available batches!')
        if x is not None:
            Input = np.concatenate([x, t], axis=1) if (td and t is not None) else x
        if MORvar is not None:
            if batch is None:
                MORarg = np.asarray(MORarg, dtype=float)
                if MORarg.shape[0] == 1:
                    MORarg = np.tile(MORarg, [dof, 1])
                Input = np.hstack([Input, MORarg])
            else:
                cols, _, _ = self._mor_columns(batch, Input.shape[0])
                Input = np.hstack([Input, cols])
        return self.engine.forward(Input).cpu().numpy().astype(np.float64).reshape(-1, 1)

    def residual(self, Input=None, tDiscIND=None, batch=None, fp64=False):
        """
        Strong PDE residual norm on `uniform_input` (or `Input`) and, when an exact solution is
        known, the normalised l2 error (VarNet.py:1599-1692):
            res = sqrt(sum(resVec^2) * prod(hVec)), err = l2Err(cEx, cApp), averaged over MOR batches.
        Returns (res, resVec, err, cApp).
        """
        if self._towers is not None:
            return self._towers.call('residual', Input, tDiscIND, batch, fp64)
        dim, PDE, fd = self.dim, self.PDE, self.fixData
        td = PDE.timeDependent
        elemSize = np.prod(fd.hVec)
        nb = fd.MORbatchNum
        if batch is not None:
            if batch > nb - 1:
                raise ValueError('requested batch number is higher than total available batches!')
            batchRange, nb = range(batch, batch + 1), 1
        else:
            batchRange = range(nb)
        noInput = Input is None
        cEx = None
        if noInput:
            Input, cEx = fd.uniform_input, fd.cEx
        elif PDE.cEx is not None:
            cEx = PDE.cEx(Input[:, :dim], Input[:, dim:dim + 1]) if td else PDE.cEx(Input[:, :dim])
        targ = [Input[:, dim:dim + 1]] if td else []
        diff_dx = fd.d_diff if noInput else PDE.d_diffFun(Input[:, :dim], *targ)
        res, err = 0., (0. if PDE.cEx is not None else None)
        resVec = cApp = None
        for b in batchRange:
            if PDE.MORvar is None:
                if noInput and fd.uniform_inpData[0] is not None:
                    diff, vel, src = fd.uniform_inpData
                else:
                    diff, vel, src = self.PDEinpData(Input)
                Inp = Input
            else:
                cols, _, inpArg = self._mor_columns(b, Input.shape[0])
                diff, vel, src = self.PDEinpData(Input, inpArg)
                Inp = np.hstack([Input, cols])
            u, r = self.engine.residual(Inp, diff, vel, src, diff_dx, fp64=fp64)
            cApp = u.cpu().numpy().astype(np.float64).reshape(-1, 1)
            resVec = r.cpu().numpy().astype(np.float64).reshape(-1, 1)
            if PDE.cEx is not None:
                err += uf.l2Err(cEx, cApp)
            res += np.sqrt(np.sum(resVec ** 2) * elemSize)
        res = res / nb
        if PDE.cEx is not None:
            err = err / nb
        return res, resVec, err, cApp
