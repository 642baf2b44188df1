"""CPU, world_size 2, gloo: the row sharding and the four exchanges of flgp_amd.pipeline
(all-gather of anchors, all-reduce of column sums / Gram partials, training-block broadcast)
reproduce the single-process result.  Stages are oracle-backed (tests/oracle_stages.py): this
tests the exchange logic, the HIP stages themselves are covered by the -m gpu parity tests."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from flgp_amd import synth  # noqa: E402
from flgp_amd.pipeline import HeatKernelPath, NystromPath, PathConfig, shard_bounds  # noqa: E402

N, D, S, R, K, M, T = 600, 3, 40, 4, 8, 50, 4.0


def _inputs(lo, hi):
    X = synth.gaussian_mixture(hi - lo, D, components=4, seed=77, row_offset=lo)
    sel = np.sort(synth.random_anchor_rows(N, S, seed=77))
    mine = sel[(sel >= lo) & (sel < hi)] - lo
    return X, np.ascontiguousarray(X[mine, :].T)


def _run(rank, world, port, kernel, gl, out):
    from oracle_stages import OracleStages
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(N, world, rank)
        X, U_loc = _inputs(lo, hi)
        path = HeatKernelPath(OracleStages())
        assert path.world == world and path.rank == rank
        Xt = torch.from_numpy(np.ascontiguousarray(X.T))
        U = path.gather_anchors(torch.from_numpy(U_loc))
        assert U.shape == (D, S)
        anchors = path.stages.anchor_prep(U)
        sizes = path.cluster_sizes(Xt, anchors)
        assert float(sizes.sum()) == N
        cfg = PathConfig(s=S, r=R, K=K, t=T, m=M, kernel=kernel, gl=gl, root=True, epsilon=0.6)
        res = path.run(Xt, U, cfg, N, lo, num_class=sizes)
        np.save(os.path.join(out, f"H_{world}_{rank}.npy"), res.H.numpy().T)          # (n_loc, m)
        np.save(os.path.join(out, f"vals_{world}_{rank}.npy"), res.values.numpy())
        if rank == 0:
            np.save(os.path.join(out, f"U_{world}.npy"), U.numpy())
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("kernel,gl", [("lae", "cluster-normalized"), ("se", "normalized"), ("lae", "rw")])
def test_two_ranks_match_one(tmp_path, kernel, gl):
    out = str(tmp_path)
    for world in (1, 2):
        mp.spawn(_run, args=(world, _free_port(), kernel, gl, out), nprocs=world, join=True)
    U1, U2 = np.load(os.path.join(out, "U_1.npy")), np.load(os.path.join(out, "U_2.npy"))
    np.testing.assert_array_equal(U1, U2)                       # exchange 1: same anchor set, same order
    H1 = np.load(os.path.join(out, "H_1_0.npy"))
    H2 = np.vstack([np.load(os.path.join(out, f"H_2_{r}.npy")) for r in range(2)])
    assert H1.shape == (N, M) and H2.shape == (N, M)
    # column sums / Gram are reduced in a different association across ranks: rounding-level only
    assert np.abs(H1 - H2).max() <= 1e-9 * np.abs(H1).max()
    v1 = np.load(os.path.join(out, "vals_1_0.npy"))
    for r in range(2):
        np.testing.assert_allclose(np.load(os.path.join(out, f"vals_2_{r}.npy")), v1, rtol=1e-11)


def _run_nystrom(rank, world, port, out):
    from oracle_stages import OracleStages
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(N, world, rank)
        X, U_loc = _inputs(lo, hi)
        path = NystromPath(OracleStages())
        U = path.gather_anchors(torch.from_numpy(U_loc))
        vals, vecs = path.run_nystrom(torch.from_numpy(np.ascontiguousarray(X.T)), U, 0.8, K)
        np.save(os.path.join(out, f"nv_{world}_{rank}.npy"), vecs.numpy().T)
        np.save(os.path.join(out, f"nl_{world}_{rank}.npy"), vals.numpy())
    finally:
        dist.destroy_process_group()


def test_nystrom_two_ranks_match_one(tmp_path):
    """SURVEY 8f-3 over shards: one anchor all-gather, then nothing -- the stacked row blocks equal the single-rank
    extension (the anchor-side eigensolve is replicated and deterministic; on the GPU the match is bit for bit,
    tests/test_gpu_parity.py::test_nystrom_row_shards_bit_identical)."""
    out = str(tmp_path)
    for world in (1, 2):
        mp.spawn(_run_nystrom, args=(world, _free_port(), out), nprocs=world, join=True)
    V1 = np.load(os.path.join(out, "nv_1_0.npy"))
    V2 = np.vstack([np.load(os.path.join(out, f"nv_2_{r}.npy")) for r in range(2)])
    assert V1.shape == (N, K)
    np.testing.assert_allclose(V2, V1, rtol=0, atol=1e-12 * np.abs(V1).max())   # numpy's GEMM may block rows differently
    for r in range(2):
        np.testing.assert_array_equal(np.load(os.path.join(out, f"nl_2_{r}.npy")), np.load(os.path.join(out, "nl_1_0.npy")))


def test_single_process_matches_oracle_pipeline():
    """No process group: HeatKernelPath degenerates to the plain path and equals the oracle's
    heat_kernel_covariance on the same inputs."""
    from oracle import flgp_oracle as O
    from oracle_stages import OracleStages
    X, U_loc = _inputs(0, N)
    path = HeatKernelPath(OracleStages())
    Xt = torch.from_numpy(np.ascontiguousarray(X.T))
    U = path.gather_anchors(torch.from_numpy(U_loc))
    anchors = path.stages.anchor_prep(U)
    sizes = path.cluster_sizes(Xt, anchors)
    cfg = PathConfig(s=S, r=R, K=K, t=T, m=M)
    res = path.run(Xt, U, cfg, N, 0, num_class=sizes)
    Ufull = np.asfortranarray(np.hstack([U.numpy().T, sizes.numpy()[:, None]]))
    Ho = O.heat_kernel_covariance(X[:M], X[M:], Ufull, R, T, K=K, method="gram")
    assert np.abs(res.H.numpy().T - Ho).max() <= 1e-9 * np.abs(Ho).max()
