"""Oracle-backed implementation of the stage interface of flgp_amd.pipeline (TEST ONLY).

Lets the CPU tests drive HeatKernelPath -- the row sharding and the four cross-rank exchanges --
under gloo, where no GPU (and therefore no HipStages) exists.  Same tensor conventions as
HipStages: column-major (n x k) matrices are torch tensors of shape (k, n)."""
import numpy as np
import torch

from oracle import flgp_oracle as O
from flgp_amd.pipeline import _WallTimer


def _np_cm(t):   # (k, n) tensor -> (n, k) Fortran array
    return np.asfortranarray(t.numpy().T)


def _t_cm(a):    # (n, k) array -> (k, n) tensor
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a).T))


class OracleStages:
    def anchor_prep(self, U):
        return dict(U=_np_cm(U), s=U.shape[1], d=U.shape[0])

    def knn(self, X, anchors, r, want_dist=False):
        res = O.knn(_np_cm(X), anchors["U"], r, output=want_dist)
        if want_dist:
            return _t_cm(res[0]), _t_cm(res[1])
        return _t_cm(res), None

    def lae(self, X, anchors, knn_idx):
        ei, ev = O.lae(_np_cm(X), anchors["U"], knn_idx.shape[0], knn_idx=_np_cm(knn_idx))
        return torch.from_numpy(ei), torch.from_numpy(ev)

    def se_weights(self, knn_idx, knn_dist, epsilon):
        ei, ev = O.se_weights(_np_cm(knn_idx), _np_cm(knn_dist), epsilon)
        return torch.from_numpy(ei), torch.from_numpy(ev)

    def csc(self, ell_idx, s):
        return dict(s=s, idx=ell_idx)

    def colsum(self, ell_idx, ell_val, s):
        return torch.from_numpy(O.colsum(ell_idx.numpy(), ell_val.numpy(), s))

    def col_scale(self, ell_idx, ell_val, colsum, num_class, mode):
        c = colsum.numpy()
        f = 1.0 / (c + 1e-9) if mode == 0 else 1.0 / np.sqrt(np.abs(c) + 1e-9)
        v = ell_val.numpy()
        v *= f[ell_idx.numpy()]
        if mode == 0 and num_class is not None:
            v *= num_class.numpy()[ell_idx.numpy()]

    def row_normalize(self, ell_val):
        v = ell_val.numpy()
        v *= (1.0 / (v.sum(1) + 1e-9))[:, None]

    def gram(self, ell_idx, ell_val, csc):
        return torch.from_numpy(O.gram(ell_idx.numpy(), ell_val.numpy(), csc["s"]))

    def sym_pack(self, G):
        g = G.numpy(); s = g.shape[0]
        return torch.from_numpy(np.concatenate([g[j, :j + 1] for j in range(s)]))     # (s, s) tensor == column-major: g[j] is column j

    def sym_unpack(self, p, G):
        s = G.shape[0]; v = p.numpy(); g = np.zeros((s, s)); o = 0
        for j in range(s):
            g[j, :j + 1] = v[o:o + j + 1]; g[:j + 1, j] = v[o:o + j + 1]; o += j + 1
        return torch.from_numpy(g)

    def eig_topk(self, G, K, tol=0.0):
        w, V = np.linalg.eigh(G.numpy())
        w = w[::-1][:K].copy(); V = V[:, ::-1][:, :K]
        V = V * np.sign(V[np.abs(V).argmax(0), np.arange(K)])[None, :]   # deterministic signs across ranks
        return torch.from_numpy(w), _t_cm(V), dict(outer_iterations=0, g_products=0, dense=True)

    def u_recover(self, ell_idx, ell_val, V, eig, scale, root, dense=None):
        K, s = V.shape
        sig = np.sqrt(np.maximum(eig.numpy(), 0.0))
        n = ell_idx.shape[0]
        vec = O.u_recover(ell_idx.numpy(), ell_val.numpy(), s, _np_cm(V), sig) / np.sqrt(float(max(n, 1))) * scale
        vals = sig if root else sig ** 2
        return torch.from_numpy(vals.copy()), _t_cm(vec)

    def hk(self, values, t, V0, V1):
        w = np.exp(-t * (1.0 - values.numpy()))
        H = (_np_cm(V0) * w[None, :]) @ _np_cm(V1).T
        return _t_cm(H)

    def nystrom(self, X, U, a2, K):
        vals, vecs = O.np_nystrom_eigenpair(_np_cm(X), _np_cm(U), a2, K)
        return torch.from_numpy(vals.copy()), _t_cm(vecs)

    def bincount(self, idx_row, s):
        return torch.bincount(idx_row.to(torch.int64), minlength=s).to(torch.float64)

    def sync(self):
        pass

    def timer(self):
        return _WallTimer()
