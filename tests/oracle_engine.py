"""
TEST-ONLY stand-in for varnet_amd.engine.VNEngine that computes with the oracle
(oracle/tangent_ref.py + oracle/tf1_graph.py) on the CPU.  It lets the host logic of
`VarNet` (data assembly, batching, tower sharding, weight rules, the training loop and the
gradient all-reduce) run in the CPU test tier, including world_size-2 gloo runs.  It is never
importable from the product package.
"""
import numpy as np
import torch

from oracle import tf1_graph as og
from oracle import tangent_ref as tr


class OracleEngine:
    def __init__(self, dim, inpDim, layerWidth, timeDependent, integNum, isSource=False,
                 integWflag=False, learning_rate=0.001, device=0, activationFun='sigmoid',
                 optimizer_name='adam', kernel=0, dtype=np.float64):
        self.torch = torch
        self.device = torch.device('cpu')
        self.dim, self.inpDim, self.layerWidth = dim, inpDim, list(layerWidth)
        self.td, self.integNum = bool(timeDependent), int(integNum)
        self.isSource, self.integWflag = bool(isSource), bool(integWflag)
        self.P = og.param_count(inpDim, layerWidth)
        self.dtype = dtype
        self.lr = learning_rate
        self.theta = np.zeros(self.P, dtype=dtype)
        self.adam = og.TF1Adam(self.P, lr=learning_rate, dtype=dtype)
        self.batches, self.bic = {}, None
        self.w = np.ones(3)
        self.gradbuf = None
        self.fe = None

    def dev(self, a, dtype=None):
        if isinstance(a, torch.Tensor):
            return a.to(torch.float64).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64)

    def init_params(self, seed=0):
        self.theta = og.glorot_init(self.inpDim, self.layerWidth, seed).astype(self.dtype)
        self.adam = og.TF1Adam(self.P, lr=self.lr, dtype=self.dtype)

    def get_params(self):
        return self.theta.astype(np.float32)

    def set_params(self, flat):
        self.theta = np.asarray(flat).astype(self.dtype)

    def export_state(self):
        # the C ABI's layout (include/varnet_hip.h, vn_state_export): int64 step, then theta, m, v as float32
        step = np.array([self.adam.t], dtype=np.int64).view(np.uint8)
        body = np.concatenate([self.theta, self.adam.m, self.adam.v]).astype(np.float32).view(np.uint8)
        return np.concatenate([step, body])

    def import_state(self, buf):
        buf = np.asarray(buf, dtype=np.uint8)
        P = self.P
        self.adam.t = int(buf[:8].view(np.int64)[0])
        a = buf[8:].view(np.float32).astype(self.dtype)
        self.theta, self.adam.m, self.adam.v = a[:P].copy(), a[P:2 * P].copy(), a[2 * P:3 * P].copy()

    def state_snapshot(self):
        self._snap = (self.theta.copy(), self.adam.m.copy(), self.adam.v.copy(), self.adam.t)

    def state_rollback(self):
        th, m, v, t = self._snap
        self.theta, self.adam.m, self.adam.v, self.adam.t = th.copy(), m.copy(), v.copy(), t

    @property
    def step(self):
        return self.adam.t

    def set_fe_table(self, N, dNt, integW=None):
        self.fe = (np.reshape(N, -1).astype(self.dtype), np.reshape(dNt, -1).astype(self.dtype),
                   None if integW is None else np.reshape(integW, -1).astype(self.dtype))

    def set_interior(self, batch, Input, gcoef, source=None, n_k=None, detJ=1.0, N_rows=None, dNt_rows=None):
        npy = lambda t: None if t is None else (t.numpy().copy() if isinstance(t, torch.Tensor) else np.array(t, dtype=float))
        dj = npy(detJ) if isinstance(detJ, torch.Tensor) or np.size(detJ) > 1 else float(np.reshape(detJ, -1)[0])
        self.batches[batch] = (npy(Input), npy(gcoef), npy(source), int(n_k), dj, npy(N_rows), npy(dNt_rows))
        getattr(self, 'bbic', {}).pop(batch, None)

    def set_bic(self, biInput, biLabel, bDof, biDimVal):
        self._bic_key = None
        self.bic = (biInput.numpy().copy(), biLabel.numpy().copy(), int(bDof), float(biDimVal))

    def set_batch_bic(self, batch, biInput=None, biLabel=None):
        if not hasattr(self, 'bbic'):
            self.bbic = {}
        if biInput is None:
            self.bbic.pop(batch, None)
        else:
            self.bbic[batch] = (biInput.numpy().copy(), biLabel.numpy().copy())

    def set_weights(self, w):
        self.w = np.array(w, dtype=float)

    def bind_grad_buffer(self):
        if self.gradbuf is None:
            self.gradbuf = torch.zeros(self.P + 4, dtype=torch.float64)
        return self.gradbuf

    def _eval(self, batch):
        Input, gcoef, src, n_k, detJ, Nr, dNtr = self.batches[batch]
        biInput, biLabel, bDof, biDimVal = self.bic
        if batch in getattr(self, 'bbic', {}):
            biInput, biLabel = self.bbic[batch]
        N, dNt, W = self.fe
        q = self.integNum
        Nrow = np.tile(N, n_k) if Nr is None else Nr.reshape(-1)
        dNtrow = np.tile(dNt, n_k) if dNtr is None else dNtr.reshape(-1)
        return tr.loss_and_grad(self.theta.astype(np.float64), self.inpDim, self.layerWidth, self.dim,
                                Input, gcoef, src if self.isSource else None, Nrow, dNtrow,
                                W if self.integWflag else None, n_k, q, detJ, biInput, biLabel, bDof, biDimVal,
                                self.w, self.td)

    def grad(self, batch=0):
        res, g = self._eval(batch)
        gb = self.bind_grad_buffer()
        gb[:self.P] = torch.as_tensor(g)
        gb[self.P:] = torch.tensor([res['loss'], res['BCloss'], res['ICloss'], res['varLoss']])

    def apply(self):
        self.theta = self.adam.step(self.theta, self.gradbuf[:self.P].numpy())

    def train_step(self, batch=0, loss_out=None):
        self.grad(batch)
        if loss_out is not None:
            loss_out[0] = self.gradbuf[self.P]
        self.apply()

    def eval_loss(self, batch=0, lossVec=False):
        res, _ = self._eval(batch)
        lv = torch.as_tensor(res['lossVec']) if lossVec else None
        return [res['loss'], res['BCloss'], res['ICloss'], res['varLoss']], lv

    def forward(self, X):
        X = X.numpy() if isinstance(X, torch.Tensor) else np.asarray(X)
        return torch.as_tensor(og.forward(self.theta, self.inpDim, self.layerWidth, torch.float64, X)[:, 0])

    def residual(self, X, diff, vel, source=None, diff_dx=None, fp64=False):
        n = np.shape(X)[0]
        src = np.zeros((n, 1)) if source is None else np.reshape(source, (n, 1))
        ddx = np.zeros((n, self.dim)) if diff_dx is None else np.reshape(diff_dx, (n, self.dim))
        u, r = og.residual(self.theta, self.inpDim, self.layerWidth, torch.float64, np.asarray(X),
                           np.reshape(diff, (n, 1)), np.reshape(vel, (n, self.dim)), src, ddx, self.dim, self.td)
        return torch.as_tensor(u[:, 0]), torch.as_tensor(r[:, 0])

    def close(self):
        pass
