"""Device-resident, row-sharded driver of the heat-kernel covariance path.

One process per GPU.  Points are partitioned by contiguous row blocks (rank g owns rows
[g*n/P, (g+1)*n/P) of X, and the same rows of idx / val / V / H); every per-point stage
(k-NN, LAE, row normalisation, U-recovery, H rows) runs on local rows only.  The exchanges
(torch.distributed: RCCL over xGMI on GPUs, gloo in the CPU tests) are exactly the four of
SURVEY.md §8(e):

  1. all-gather of the anchor sets (+ all-reduce of the 1-NN cluster counts),
  2. all-reduce(sum) of the column sums of Z  (8 s bytes, twice),
  3. all-reduce(sum) of the Gram partials     (upper triangle, 4 s(s+1) bytes -- the one real exchange);
     the s x s eigensolve then runs replicated, with no further communication,
  4. sum-all-reduce of the zero-padded training block V[0:m] (m x K): a broadcast from
     its owners that needs no ownership bookkeeping.

torch is plumbing here (device memory, streams, the process group); all arithmetic happens
behind the C ABI of libflgp_hip.so through a ``stages`` object.  ``HipStages`` is the only
implementation in the package -- there is no CPU fallback; the CPU sharding tests inject
their own oracle-backed stages to exercise the exchange logic.

Column-major convention: an (n x k) column-major matrix is held as a contiguous torch tensor
of shape (k, n), i.e. ``t[j, i]`` is element (i, j).  ELL arrays are (n, r) row-major.
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass, field

import torch

from . import _lib

GL_CODES = {"rw": 0, "normalized": 1, "cluster-normalized": 2}


def shard_bounds(n: int, world: int, rank: int):
    """Rows [lo, hi) owned by ``rank``: contiguous blocks, sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class HipStages:
    """The device stages behind the C ABI (include/flgp_hip.h), on torch CUDA tensors."""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("HipStages needs a GPU: the HIP library is the only implementation of this path")
        self.L = _lib.lib()
        torch.cuda.set_device(self.device)
        _lib.check(self.L.flgp_set_device(self.device.index or 0))

    # -- helpers
    def _st(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def empty(self, shape, dtype=torch.float64):
        return torch.empty(shape, dtype=dtype, device=self.device)

    # -- stages
    def anchor_prep(self, U):  # U: (d, s) == column-major s x d
        d, s = U.shape
        dpad = self.L.flgp_dev_anchor_dpad(d)
        rows = self.L.flgp_dev_anchor_rows(s)
        Ut = self.empty((rows, dpad)); uu = self.empty((rows,))
        _lib.check(self.L.flgp_dev_anchor_prep(self._st(), U.data_ptr(), s, s, d, Ut.data_ptr(), uu.data_ptr()))
        return dict(Ut=Ut, uu=uu, s=s, d=d)

    def knn(self, X, anchors, r, want_dist=False):  # X: (d, n_loc)
        d, n = X.shape
        idx = self.empty((r, n), torch.int32)
        dist = self.empty((r, n)) if want_dist else None
        _lib.check(self.L.flgp_dev_knn(self._st(), X.data_ptr(), n, n, d, anchors["Ut"].data_ptr(), anchors["uu"].data_ptr(),
                                       anchors["s"], r, idx.data_ptr(), dist.data_ptr() if want_dist else None, n))
        return idx, dist

    def lae(self, X, anchors, knn_idx):
        d, n = X.shape
        r = knn_idx.shape[0]
        ei = self.empty((n, r), torch.int32); ev = self.empty((n, r))
        _lib.check(self.L.flgp_dev_lae(self._st(), X.data_ptr(), n, n, d, anchors["Ut"].data_ptr(), anchors["s"], r,
                                       knn_idx.data_ptr(), n, ei.data_ptr(), ev.data_ptr()))
        return ei, ev

    def se_weights(self, knn_idx, knn_dist, epsilon):
        r, n = knn_idx.shape
        ei = self.empty((n, r), torch.int32); ev = self.empty((n, r))
        _lib.check(self.L.flgp_dev_se_weights(self._st(), knn_idx.data_ptr(), knn_dist.data_ptr(), n, n, r, float(epsilon),
                                              ei.data_ptr(), ev.data_ptr()))
        return ei, ev

    def csc(self, ell_idx, s):
        n, r = ell_idx.shape
        wb = self.L.flgp_dev_csc_workspace(n, s, r)
        work = self.empty((wb // 8 + 1,))
        colptr = self.empty((s + 1,), torch.int32); pos = self.empty((max(n * r, 1),), torch.int32)
        _lib.check(self.L.flgp_dev_csc_build(self._st(), ell_idx.data_ptr(), n, s, r, colptr.data_ptr(), pos.data_ptr(),
                                             work.data_ptr(), wb))
        return dict(colptr=colptr, pos=pos, s=s)

    def colsum(self, ell_idx, ell_val, s):
        n, r = ell_idx.shape
        out = self.empty((s,))
        wb = self.L.flgp_dev_colsum_workspace(n, s)
        work = self.empty((wb // 8 + 1,))
        _lib.check(self.L.flgp_dev_colsum(self._st(), ell_idx.data_ptr(), ell_val.data_ptr(), n, r, s, out.data_ptr(),
                                          work.data_ptr(), wb))
        return out

    def col_scale(self, ell_idx, ell_val, colsum, num_class, mode):
        n, r = ell_idx.shape
        _lib.check(self.L.flgp_dev_col_scale(self._st(), ell_idx.data_ptr(), ell_val.data_ptr(), n, r, colsum.data_ptr(),
                                             num_class.data_ptr() if num_class is not None else None, mode))

    def row_normalize(self, ell_val):
        n, r = ell_val.shape
        _lib.check(self.L.flgp_dev_row_normalize(self._st(), ell_val.data_ptr(), n, r))

    def col_scale_row_normalize(self, ell_idx, ell_val, colsum, num_class):
        """col_scale(mode 0) and row_normalize in one pass over the values (the same bits)."""
        n, r = ell_idx.shape
        _lib.check(self.L.flgp_dev_col_scale_row_normalize(self._st(), ell_idx.data_ptr(), ell_val.data_ptr(), n, r, colsum.data_ptr(),
                                                           num_class.data_ptr() if num_class is not None else None))

    def gram(self, ell_idx, ell_val, csc):
        n, r = ell_idx.shape
        s = csc["s"]
        G = self.empty((s, s))
        _lib.check(self.L.flgp_dev_gram(self._st(), ell_idx.data_ptr(), ell_val.data_ptr(), n, s, r, csc["colptr"].data_ptr(),
                                        csc["pos"].data_ptr(), G.data_ptr(), s))
        return G

    def sym_pack(self, G):
        s = G.shape[0]
        p = self.empty((s * (s + 1) // 2,))
        _lib.check(self.L.flgp_dev_sym_pack(self._st(), G.data_ptr(), s, s, p.data_ptr()))
        return p

    def sym_unpack(self, p, G):
        s = G.shape[0]
        _lib.check(self.L.flgp_dev_sym_unpack(self._st(), p.data_ptr(), s, G.data_ptr(), s))
        return G

    def eig_topk(self, G, K, tol=0.0):
        s = G.shape[0]
        wb = self.L.flgp_dev_eig_workspace(s, K)
        work = self.empty((wb // 8 + 1,))
        eig = self.empty((K,)); V = self.empty((K, s))
        import ctypes
        info = (ctypes.c_int * 4)()
        _lib.check(self.L.flgp_dev_eig_topk(self._st(), G.data_ptr(), s, s, K, float(tol), eig.data_ptr(), V.data_ptr(), s,
                                            work.data_ptr(), wb, ctypes.addressof(info)))
        return eig, V, dict(outer_iterations=info[0], g_products=info[1], dense=bool(info[2]),
                            newton_schulz_orths=info[3] // 1000, jacobi_orths=info[3] % 1000)

    def u_recover(self, ell_idx, ell_val, V, eig, scale, root, dense=None):
        """dense: did the full decomposition deliver `eig` (eig_topk's info["dense"])?  Default: K == s."""
        n, r = ell_idx.shape
        K, s = V.shape
        if dense is None:
            dense = K >= s
        values = self.empty((K,)); vectors = self.empty((K, max(n, 1)))
        work = self.empty((self.L.flgp_dev_u_recover_workspace(s, K) // 8 + 1,))
        _lib.check(self.L.flgp_dev_spectrum_usable_route(self._st(), eig.data_ptr(), K, int(bool(dense))))   # sigma_K resolved, or the reason why not
        _lib.check(self.L.flgp_dev_u_recover(self._st(), ell_idx.data_ptr(), ell_val.data_ptr(), n, r, V.data_ptr(), s, s,
                                             eig.data_ptr(), K, float(scale), int(bool(root)), vectors.data_ptr(), max(n, 1),
                                             values.data_ptr(), work.data_ptr()))
        return values, vectors

    def hk(self, values, t, V0, V1):
        K, n0 = V0.shape
        n1 = V1.shape[1]
        H = self.empty((n1, n0))
        wb = self.L.flgp_dev_hk_workspace(n0, n1, K, 0)
        work = self.empty((wb // 8 + 1,))
        _lib.check(self.L.flgp_dev_hk(self._st(), values.data_ptr(), K, float(t), V0.data_ptr(), n0, None, 0, n0,
                                      V1.data_ptr(), n1, None, 0, n1, H.data_ptr(), n0, work.data_ptr()))
        return H

    def nystrom(self, X, U, a2, K):  # X: (d, n_loc), U: (d, s) -> values (K,), vectors (K, n_loc)
        d, n = X.shape
        s = U.shape[1]
        values = self.empty((K,)); vectors = self.empty((K, max(n, 1)))
        if n:
            _lib.check(self.L.flgp_dev_nystrom_eigenpair(self._st(), X.data_ptr(), n, n, d, U.data_ptr(), s, s, float(a2), int(K),
                                                         values.data_ptr(), vectors.data_ptr(), n))
        return values, vectors

    # -- plumbing used by the driver (no arithmetic of the path)
    def bincount(self, idx_row, s):
        return torch.bincount(idx_row.to(torch.int64), minlength=s).to(torch.float64)

    def sync(self):
        torch.cuda.synchronize(self.device)

    def timer(self):
        return _CudaTimer(self.device)

    def side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side


class _CudaTimer:
    """HIP-event timer on the stream the kernels are launched on (torch's current stream)."""

    def __init__(self, device):
        self.device = device
        self.marks = []

    def mark(self, name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(self.device))
        self.marks.append((name, ev))

    def result(self):
        torch.cuda.synchronize(self.device)
        out = {}
        for (n0, e0), (n1, e1) in zip(self.marks[:-1], self.marks[1:]):
            out[n1] = out.get(n1, 0.0) + e0.elapsed_time(e1)
        return out


class _WallTimer:
    def __init__(self):
        self.marks = []

    def mark(self, name):
        self.marks.append((name, time.perf_counter()))

    def result(self):
        out = {}
        for (n0, t0), (n1, t1) in zip(self.marks[:-1], self.marks[1:]):
            out[n1] = out.get(n1, 0.0) + (t1 - t0) * 1e3
        return out


@dataclass
class PathConfig:
    s: int
    r: int
    K: int
    t: float
    m: int                     # training rows = global rows [0, m)
    kernel: str = "lae"
    gl: str = "cluster-normalized"
    root: bool = True
    epsilon: float = 0.1


@dataclass
class PathResult:
    values: torch.Tensor               # (K,)
    vectors: torch.Tensor              # (K, n_loc): local rows of V = U sqrt(n)
    H: torch.Tensor                    # (m, n_loc): local rows of the n x m covariance, column-major
    stage_ms: dict = field(default_factory=dict)
    eig_info: dict = field(default_factory=dict)
    ell_idx: torch.Tensor = None
    ell_val: torch.Tensor = None
    knn_idx: torch.Tensor = None
    G: torch.Tensor = None              # (s, s) Gram matrix the eigensolver saw (keep=True)


class HeatKernelPath:
    """k-NN -> LAE/SE similarity -> graph Laplacian -> top-K spectrum -> heat-kernel covariance
    on the local row block, with the cross-rank exchanges described in the module docstring."""

    def __init__(self, stages, group=None):
        self.stages = stages
        self.dist = None
        self.group = group
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.dist = dist
        except Exception:  # pragma: no cover
            self.dist = None
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.rank = self.dist.get_rank(group) if self.dist else 0

    # -- collectives (no-ops on one rank)
    def _all_reduce(self, t):
        if self.dist and self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def gather_anchors(self, U_local):
        """Exchange 1: all-gather of the per-rank anchor sets.  U_local: (d, s_loc) column-major
        s_loc x d block; ranks may contribute different counts (padded to the maximum)."""
        if not self.dist or self.world == 1:
            return U_local.contiguous()
        d, s_loc = U_local.shape
        cnt = torch.tensor([s_loc], dtype=torch.int64, device=U_local.device)
        cnts = [torch.zeros_like(cnt) for _ in range(self.world)]
        self.dist.all_gather(cnts, cnt, group=self.group)
        cnts = [int(c.item()) for c in cnts]
        smax = max(cnts)
        pad = torch.zeros((smax, d), dtype=U_local.dtype, device=U_local.device)
        pad[:s_loc] = U_local.t()
        parts = [torch.zeros_like(pad) for _ in range(self.world)]
        self.dist.all_gather(parts, pad, group=self.group)
        rows = torch.cat([p[:c] for p, c in zip(parts, cnts)], dim=0)  # (s, d) row-major
        return rows.t().contiguous()                                   # (d, s)

    def cluster_sizes(self, X_loc, anchors):
        """1-NN assignment counts of every anchor over ALL points (semantics of the reference's
        minibatchkmeans branch, src/Utils.cpp:59-62): local k-NN with r = 1, then all-reduce."""
        idx, _ = self.stages.knn(X_loc, anchors, 1)
        counts = self.stages.bincount(idx[0], anchors["s"])
        return self._all_reduce(counts)

    def run(self, X_loc, U, cfg: PathConfig, n_global: int, row_lo: int, num_class=None, keep=False) -> PathResult:
        """X_loc: (d, n_loc) local rows [row_lo, row_lo + n_loc) of X_all.  U: (d, s) anchors
        (already exchanged).  num_class: (s,) cluster sizes or None."""
        S = self.stages
        if cfg.gl not in GL_CODES:
            raise _lib.FlgpError(-3, "Error: the type of graph Laplacian is not supported!")
        if cfg.kernel not in ("lae", "se"):
            raise _lib.FlgpError(-3, "The kernel type is not supported!")
        gl = GL_CODES[cfg.gl]
        if gl == 2 and num_class is None:
            raise _lib.FlgpError(-1, 'gl="cluster-normalized" needs the cluster sizes')
        d, n_loc = X_loc.shape
        s = U.shape[1]
        tm = S.timer()
        tm.mark("start")
        anchors = S.anchor_prep(U)
        # k1 + k2
        knn_idx, knn_dist = S.knn(X_loc, anchors, cfg.r, want_dist=(cfg.kernel == "se"))
        tm.mark("knn")
        # k3 + k4
        if cfg.kernel == "lae":
            ell_idx, ell_val = S.lae(X_loc, anchors, knn_idx)
        else:
            ell_idx, ell_val = S.se_weights(knn_idx, knn_dist, cfg.epsilon)
        tm.mark("similarity")
        # k5: graph Laplacian (reference src/Utils.cpp:195-212).  The CSC view depends on the pattern alone and the scalings
        # only touch the values: the view is built on a second stream beside them and joined before the Gram kernel.
        side = S.side_stream() if hasattr(S, "side_stream") else None       # (a stage set without streams builds it in line)
        if side is not None:
            main = torch.cuda.current_stream(S.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                csc = S.csc(ell_idx, s)
        else:
            csc = S.csc(ell_idx, s)
        if gl != 0:
            c = self._all_reduce(S.colsum(ell_idx, ell_val, s))                # exchange 2a
            if hasattr(S, "col_scale_row_normalize"):                          # one pass over the values instead of two
                S.col_scale_row_normalize(ell_idx, ell_val, c, num_class if gl == 2 else None)
            else:
                S.col_scale(ell_idx, ell_val, c, num_class if gl == 2 else None, 0)
                S.row_normalize(ell_val)
        else:
            S.row_normalize(ell_val)
        # spectrum scaling (src/Spectrum.cpp:149-150)
        c2 = self._all_reduce(S.colsum(ell_idx, ell_val, s))                   # exchange 2b
        S.col_scale(ell_idx, ell_val, c2, None, 1)
        tm.mark("laplacian")
        if side is not None:
            main.wait_stream(side)
            for tns in (csc["colptr"], csc["pos"]):
                tns.record_stream(main)
        tm.mark("csc")                                                        # (what of the CSC build the scalings did not cover)
        # k6: Gram, replicated top-K eigensolve
        G = S.gram(ell_idx, ell_val, csc)
        if self.dist and self.world > 1:                                      # exchange 3: the upper triangle only
            G = S.sym_unpack(self._all_reduce(S.sym_pack(G)), G)
        tm.mark("gram")
        K = s if cfg.K < 0 else cfg.K
        eig, V, info = S.eig_topk(G, K)
        tm.mark("eig")
        values, vectors = S.u_recover(ell_idx, ell_val, V, eig, math.sqrt(float(n_global)), cfg.root, dense=info.get("dense"))
        tm.mark("u_recover")
        # k7: heat kernel against the training block V[0:m]
        m = cfg.m
        V1 = torch.zeros((K, m), dtype=vectors.dtype, device=vectors.device)
        lo, hi = row_lo, row_lo + n_loc
        if lo < m:
            cnt = min(hi, m) - lo
            V1[:, lo:lo + cnt] = vectors[:, :cnt]
        self._all_reduce(V1)                                                  # exchange 4
        H = S.hk(values, cfg.t, vectors[:, :n_loc].contiguous() if vectors.shape[1] != n_loc else vectors, V1)
        tm.mark("heat_kernel")
        res = PathResult(values=values, vectors=vectors, H=H, stage_ms=tm.result(), eig_info=info)
        if keep:
            res.ell_idx, res.ell_val, res.knn_idx, res.G = ell_idx, ell_val, knn_idx, G
        return res


class NystromPath(HeatKernelPath):
    """The Nystrom-extension spectrum (reference src/Fit.cpp:244-289) over row shards.  The extension is
    row-local and the anchor side (s x s similarity, its top-K eigenpairs) is a deterministic function of the anchors
    alone, so every rank computes it for itself: after the anchor all-gather there is no exchange at all, and the
    stacked per-rank ``vectors`` are, bit for bit, the single-rank result."""

    def run_nystrom(self, X_loc, U, a2: float, K: int):
        """X_loc: (d, n_loc) local rows; U: (d, s) anchors as returned by :meth:`gather_anchors`.
        Returns (values (K,), vectors (K, n_loc))."""
        return self.stages.nystrom(X_loc, U, a2, K)
