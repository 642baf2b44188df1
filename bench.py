"""bench.py -- points/sec through k-NN -> Laplacian -> truncated SVD -> heat-kernel covariance.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 the driver
launches it under ``torch.distributed.run`` (one rank per GPU, RCCL).  Rank 0 prints ONE
JSON line.

Workload (BASELINE.json ``configs[2]`` / ``configs[3]``): synthetic 16-component Gaussian
mixture, n = 1e6 points, d = 16, s = 5000 anchors (seeded random rows, 1-NN cluster counts),
r = 10, K = 200, kernel = "lae", gl = "cluster-normalized", root = TRUE, t = 10, m = 1000
training rows; at N GPUs the same n = 1e6 rows are sharded by row blocks (strong scaling).
A step is one pass of the whole path over the point cloud with X and the anchors already
resident in HBM and H (n x m) left in HBM.  Anchors and cluster sizes are inputs of the path
(subsample_cpp is outside it), so they are prepared before the timed region.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from flgp_amd import _lib, synth  # noqa: E402
from flgp_amd.pipeline import HeatKernelPath, HipStages, PathConfig, shard_bounds  # noqa: E402

PEAK_F64_TFLOPS = 78.6     # MI355X fp64 vector = fp64 matrix dense peak (vendor figure)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=16)
    ap.add_argument("--s", type=int, default=5000)
    ap.add_argument("--r", type=int, default=10)
    ap.add_argument("--K", type=int, default=200)
    ap.add_argument("--m", type=int, default=1000)
    ap.add_argument("--t", type=float, default=10.0)
    ap.add_argument("--cpu-rows", type=int, default=100000, help="rows of the bounded CPU-baseline sample (BASELINE.md: an n = 1e5 slice)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["lae", "se_grid"], default="lae",
                    help="lae: the headline path (BASELINE configs[2], the metric of BASELINE.json).  se_grid: the spectrum part of "
                         "fit_se_*_gp (src/Fit.cpp:127-178) at the same size -- one k-NN with distances, then TEN spectra for the "
                         "bandwidths a2s = exp(seq(log 0.1, log 10, length 10)) of R/Fit.R:128-130; a second JSON line of its own, N = 1 only")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--tune", action="append", default=[], help="key=value tuning knobs (experiments)")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not record per-kernel HIP events (no roofline object)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-boundary (PCIe-inclusive) timing of the C entry point")
    ap.add_argument("--driver", choices=["auto", "c", "python"], default="auto",
                    help="who drives the sharded path: the C entry point flgp_dev_heat_kernel_covariance_sharded with an RCCL "
                         "flgp_comm (default for N > 1), or flgp_amd/pipeline.py over torch.distributed (default for N = 1: it times the stages)")
    return ap.parse_args()


def prof_query(L, name):
    c = ctypes.c_int(0); ms = ctypes.c_double(0.0); w = ctypes.c_double(0.0)
    L.flgp_prof_query(name.encode(), ctypes.addressof(c), ctypes.addressof(ms), ctypes.addressof(w))
    return c.value, ms.value, w.value


def cpu_baseline(args, X_rows, U_np, sizes_np):
    """The oracle (kind "port") timed on this box's host cores on a bounded sample: the per-point
    stages on the first ``cpu_rows`` rows, the s x s top-K eigen-stage at full s and K on that
    sample's similarity matrix.  points/sec is extrapolated to the full n for the per-point
    stages (linear in n) with the eigen-stage counted once (independent of n)."""
    from oracle import flgp_oracle as O
    nc = X_rows.shape[0]
    t = {}
    t0 = time.perf_counter()
    kidx = O.knn(X_rows, U_np, args.r)
    t["knn"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    ei, ev = O.lae(X_rows, U_np, args.r, knn_idx=kidx)
    t["lae"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    zn = O.graph_laplacian(ei, ev, args.s, "cluster-normalized", sizes_np)
    av, _ = O.scale_A(ei, zn, args.s)
    t["laplacian"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    vals, Uu = O.truncated_svd(ei, av, args.s, args.K, method="svds")
    t["truncated_svd"] = time.perf_counter() - t0
    vec = np.asfortranarray(Uu * np.sqrt(float(nc)))
    t0 = time.perf_counter()
    mm = min(args.m, nc)
    O.hk_from_spectrum(np.sqrt(vals), vec, args.K, args.t, np.arange(nc, dtype=np.int32), np.arange(mm, dtype=np.int32))
    t["heat_kernel"] = time.perf_counter() - t0 if mm == args.m else (time.perf_counter() - t0) * args.m / mm
    scale = args.n / float(nc)
    # svds = Lanczos on the implicit operator: its sparse mat-vecs scale with n, the restart algebra does not;
    # counted unscaled here, i.e. in the CPU's favour
    total = scale * (t["knn"] + t["lae"] + t["laplacian"] + t["heat_kernel"]) + t["truncated_svd"]
    return {
        "value": args.n / total, "unit": "points/s", "cores": int(O.threads()), "kind": "port",
        "sample": f"first {nc} of {args.n} rows for k-NN/LAE/Laplacian/heat-kernel (scaled x{scale:.0f}), "
                  f"ARPACK svds at full s={args.s}, K={args.K} on that sample (counted once, unscaled); "
                  f"{int(O.threads())} OpenMP threads for the per-point stages, scipy's ARPACK + BLAS for svds",
        "stage_seconds_on_sample": {k: round(v, 4) for k, v in t.items()},
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs one process per GPU: launch with "
                         f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library is the only implementation of this path")
    # rehearsal knobs (one-GPU box): FLGP_FORCE_DEVICE pins every rank to one card and
    # FLGP_DIST_BACKEND=gloo replaces RCCL, which refuses two ranks on the same device
    if "FLGP_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["FLGP_FORCE_DEVICE"])
    backend = os.environ.get("FLGP_DIST_BACKEND", "nccl")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        # first collective right away (before any kernel of the path): RCCL builds its rings here, and the count of
        # ranks that answered goes into the JSON line
        seen = torch.ones(1, dtype=torch.int64, device=device)
        dist.all_reduce(seen)
        ranks_seen = int(seen.item())
    else:
        ranks_seen = 1
    stages = HipStages(device)
    path = HeatKernelPath(stages)
    L = _lib.lib()
    for kv in args.tune:
        k, v = kv.split("=")
        L.flgp_set_tuning(k.encode(), int(v))

    # ---- N > 1: the exchanges go through the C ABI's communicator table (RCCL, one rank per process; the 128-byte id
    #      travels over the process group that the launcher has set up).  Any failure falls back to the Python driver.
    comm = None
    driver_note = None
    want_c = args.driver == "c" or (args.driver == "auto" and world > 1 and backend == "nccl")
    if want_c and world > 1:
        try:
            import torch.distributed as dist
            idbuf = (ctypes.c_char * 128)()
            id_ok = 1
            if rank == 0 and L.flgp_comm_rccl_unique_id(idbuf) != 0:
                id_ok = 0            # (say so in the broadcast: the other ranks are waiting in it)
            idt = torch.frombuffer(bytearray(bytes(idbuf) + bytes([id_ok])), dtype=torch.uint8).to(device)
            dist.broadcast(idt, src=0)
            raw = bytes(idt.cpu().numpy().tobytes())
            if raw[128] != 1:
                raise RuntimeError("rank 0 could not create an RCCL id: " + L.flgp_last_error().decode("utf-8", "replace"))
            cbuf = (ctypes.c_char * 128).from_buffer_copy(raw[:128])
            out_c = (ctypes.c_void_p * 1)()
            _lib.check(L.flgp_comm_rccl_init_rank(world, rank, cbuf, out_c))
            comm = out_c[0]
            probe = torch.ones(4, dtype=torch.float64, device=device)
            _lib.check(L.flgp_comm_all_reduce_sum(comm, probe.data_ptr(), 4, torch.cuda.current_stream(device).cuda_stream))
            torch.cuda.synchronize(device)
            if float(probe[0].item()) != float(world):
                raise RuntimeError("the C communicator's all-reduce of ones gave %r, expected %d" % (float(probe[0].item()), world))
        except Exception as e:      # noqa: BLE001
            driver_note = "C driver unavailable (%r): Python driver over torch.distributed instead" % (e,)
            comm = None
        # every rank must take the same road
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int64, device=device)
        torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
        if int(ok.item()) == 0 and comm is not None:
            L.flgp_comm_destroy(comm); comm = None
            driver_note = "C driver unavailable on another rank: Python driver over torch.distributed instead"
    use_c = comm is not None or (args.driver == "c" and world == 1)

    n, d, s = args.n, args.d, args.s
    lo, hi = shard_bounds(n, world, rank)
    n_loc = hi - lo
    # ---- inputs: the local row block of the synthetic cloud, resident in HBM
    X_np = synth.gaussian_mixture(n_loc, d, row_offset=lo)
    X_loc = torch.from_numpy(np.ascontiguousarray(X_np.T)).to(device)       # (d, n_loc) == column-major n_loc x d
    # anchors: a global seeded row selection; every rank contributes the rows it owns (exchange 1)
    sel = np.sort(synth.random_anchor_rows(n, s))
    mine = sel[(sel >= lo) & (sel < hi)] - lo
    U_local = torch.from_numpy(np.ascontiguousarray(X_np[mine, :].T)).to(device)
    cfg = PathConfig(s=s, r=args.r, K=args.K, t=args.t, m=args.m)
    if use_c:
        st_ = lambda: torch.cuda.current_stream(device).cuda_stream     # noqa: E731
        U = torch.empty((d, s), dtype=torch.float64, device=device)
        _lib.check(L.flgp_dev_gather_anchors(st_(), comm, U_local.data_ptr(), U_local.shape[1], d, U.data_ptr(), s))
        num_class = torch.empty(s, dtype=torch.float64, device=device)
        _lib.check(L.flgp_dev_cluster_sizes(st_(), comm, X_loc.data_ptr(), n_loc, n_loc, d, U.data_ptr(), s, s, num_class.data_ptr()))
    else:
        U = path.gather_anchors(U_local)
        anchors = stages.anchor_prep(U)
        num_class = path.cluster_sizes(X_loc, anchors)                           # 1-NN counts over all ranks
    assert U.shape == (d, s)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    if args.workload == "se_grid":
        if world != 1:
            raise SystemExit("--workload se_grid is a one-GPU measurement")
        a2s = np.exp(np.linspace(np.log(0.1), np.log(10.0), 10))
        vals = torch.empty((10, args.K), dtype=torch.float64, device=device)
        iters = (ctypes.c_int * 10)()
        mean = ctypes.c_double(0.0)

        def grid(par):
            _lib.check(L.flgp_dev_se_spectrum_grid(torch.cuda.current_stream(device).cuda_stream, X_loc.data_ptr(), n, n, d, U.data_ptr(), s, s,
                                                   num_class.data_ptr(), args.r, args.K, a2s.ctypes.data, 10, b"cluster-normalized", 1,
                                                   vals.data_ptr(), None, ctypes.addressof(mean), par, ctypes.addressof(iters)))
        res_t = {}
        for par in (10, 1):                 # ten spectra at once (one host thread + stream each) vs one after the other
            for _ in range(max(1, args.warmup)):
                grid(par)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                grid(par)
            torch.cuda.synchronize(device)
            res_t[par] = (time.perf_counter() - t0) * 1e3 / args.steps
        out = {
            "metric": "points/sec through k-NN(with distances) -> 10 x [SE weights -> Laplacian -> trunc-SVD -> U] (fit_se_* bandwidth grid), n=1e6 d=16 K=200",
            "value": n / (res_t[10] * 1e-3), "unit": "points/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res_t[10], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"SE bandwidth grid, Gaussian-mixture n={n} d={d} s={s} r={args.r} K={args.K}, ten bandwidths a2 = 0.1..10 "
                                   f"(R/Fit.R:128-130), gl=cluster-normalized root=TRUE; eigenvectors (n x K per bandwidth) computed and left in HBM",
                       "spectra_in_flight": 10, "outer_iterations_per_bandwidth": [int(x) for x in iters],
                       "distances_mean": mean.value, "values_top_K_th": [[float(vals[i, 0]), float(vals[i, args.K - 1])] for i in range(10)]},
            "ms_per_step_sequential": res_t[1],
            "note": "ms_per_step: the ten spectra run concurrently (flgp_dev_se_spectrum_grid, max_parallel = 10); ms_per_step_sequential: one "
                    "after the other (max_parallel = 1).  Not the BASELINE.json metric: no roofline / cpu_baseline objects on this line.",
        }
        print(json.dumps(out))
        return

    class _CRes:
        stage_ms = {}
        eig_info = {}

    def step():
        if not use_c:
            return path.run(X_loc, U, cfg, n, lo, num_class=num_class)
        res = _CRes()
        res.H = torch.empty((args.m, n_loc), dtype=torch.float64, device=device)
        info = (ctypes.c_int * 4)()
        _lib.check(L.flgp_dev_heat_kernel_covariance_sharded(
            torch.cuda.current_stream(device).cuda_stream, comm, X_loc.data_ptr(), n_loc, n_loc, d, n, lo, U.data_ptr(), s, s,
            num_class.data_ptr(), args.m, args.r, args.t, args.K, b"lae", b"cluster-normalized", 1, 0.1, res.H.data_ptr(), n_loc,
            None, None, 0, ctypes.addressof(info)))
        res.eig_info = dict(outer_iterations=info[0], g_products=info[1], dense=bool(info[2]),
                            newton_schulz_orths=info[3] // 1000, jacobi_orths=info[3] % 1000)
        return res

    res = None
    for _ in range(args.warmup):
        res = step()
        del res
    barrier()
    L.flgp_prof_reset()
    L.flgp_prof_enable(0 if args.no_kernel_events else 1)   # 1: HIP events around the dominant kernel only
    stage_acc = {}
    t0 = time.perf_counter()
    for it in range(args.steps):
        res = step()
        for k, v in res.stage_ms.items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
        if it != args.steps - 1:
            del res
    barrier()
    elapsed = time.perf_counter() - t0
    L.flgp_prof_enable(0)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed * 1e3 / args.steps

    # ---- the dominant kernel, from the timed region (HIP events on the launch stream)
    roof = None
    c, ms, w = prof_query(L, "hk_panel_kernel")
    if c:
        ach = w / (ms * 1e-3) / 1e12
        roof = {"kernel": "hk_panel_kernel (v_mfma_f64_16x16x4_f64; the H = V diag(e^-t(1-lambda)) V^T contraction)",
                "bound": "mfma", "achieved": ach,
                "peak": PEAK_F64_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F64_TFLOPS, "traffic": None,
                "launches_per_step": c / args.steps, "avg_launch_ms": ms / c,
                "algorithmic_flops_per_launch": w / c}
    # ---- every instrumented kernel, from ONE extra step outside the timed region (recording events
    #      around ~600 launches per step costs ~10 ms per step, so it is kept out of `value`)
    kernels = {}
    if not args.no_kernel_events:
        del res
        L.flgp_prof_reset()
        L.flgp_prof_enable(2)
        res = step()
        barrier()
        L.flgp_prof_enable(0)
        for name in ["hk_panel_kernel", "gemm_f64_kernel", "gemm_large", "gemm_medium", "gemm_small", "bsg_gemm_kernel", "bsg_pre_kernel",
                     "small_gemm_kernel", "knn_kernel", "lae_kernel", "gram_kernel", "u_recover_kernel", "csc_build",
                     "colsum_kernel", "jacobi_eig", "jacobi_refine"]:
            c2, ms2, w2 = prof_query(L, name)
            if c2:
                kernels[name] = {"launches": c2, "ms": ms2, "avg_launch_ms": ms2 / c2, "work": w2}
    headline = (n, d, s, args.r, args.K, args.m) == (1_000_000, 16, 5000, 10, 200, 1000)   # the shape BASELINE.json quotes
    if roof and not (headline and world == 1):
        roof["traffic"] = None      # the committed counter passes are of the headline shape on one GPU only
    elif roof:
        try:   # HBM bytes per launch of that kernel, from the committed rocprofv3 PMC passes (profiles/)
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["hk_panel_kernel"]
            roof["traffic"] = pm["hbm_bytes_per_launch"]
            roof["traffic_source"] = pm["source"]
        except Exception:
            roof["traffic"] = None
    out = {
        "metric": "points/sec through k-NN->Laplacian->trunc-SVD->heat-cov, n=1e6 d=16 K=200" if headline else
                  f"points/sec through k-NN->Laplacian->trunc-SVD->heat-cov, n={n} d={d} K={args.K} (NOT the BASELINE.json shape)",
        "value": n / (ms_per_step * 1e-3), "unit": "points/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"Gaussian-mixture n={n} d={d} s={s} r={args.r} K={args.K} m={args.m} t={args.t} "
                               f"kernel=lae gl=cluster-normalized root=TRUE" + (f" (BASELINE configs[{2 if world == 1 else 3}])" if headline else ""),
                   "parallelism": f"row-sharded x{world}", "rows_per_gpu": n_loc, "ranks_seen": ranks_seen,
                   "driver": ("C ABI: flgp_dev_heat_kernel_covariance_sharded, exchanges through flgp_comm (RCCL)" if use_c else
                              "flgp_amd/pipeline.py (stage by stage through the C ABI" + (", exchanges over torch.distributed)" if world > 1 else ")")),
                   "driver_note": driver_note, "eig": res.eig_info},
        "stage_ms_per_step": {k: v / args.steps for k, v in stage_acc.items()},
        "kernels_diagnostic_step": kernels,
    }
    if roof:
        out["roofline"] = roof
    if rank == 0 and world == 1 and not args.no_e2e:
        # what R sees: the C entry point with host buffers in and out (X up, H = n x m down through the pipelined
        # copy).  Reported beside ms_per_step, never inside `value`.
        try:
            del res
            torch.cuda.empty_cache()
            U_h = np.asfortranarray(np.column_stack([np.ascontiguousarray(U.t().cpu().numpy()), num_class.cpu().numpy()]))
            X_h = np.asfortranarray(X_np)
            best = None
            for _ in range(3):
                H_h = np.empty((n, args.m), order="F")
                t0 = time.perf_counter()
                _lib.check(L.flgp_heat_kernel_covariance(X_h.ctypes.data, n, args.m, d, U_h.ctypes.data, s, d + 1, args.r, args.t,
                                                         args.K, b"lae", b"cluster-normalized", 1, 0.1, H_h.ctypes.data))
                dt = (time.perf_counter() - t0) * 1e3
                best = dt if best is None else min(best, dt)
                del H_h
            out["t_e2e_ms"] = best
            out["t_e2e_note"] = ("flgp_heat_kernel_covariance, host pointers in / out (X %.0f MB up, H %.1f GB down, pageable), "
                                 "best of 3 calls" % (X_h.nbytes / 1e6, n * args.m * 8 / 1e9))
        except Exception as e:  # pragma: no cover
            out["t_e2e_ms"] = None
            out["t_e2e_note"] = "failed: %r" % (e,)
    rehearse_rank_e2e = world == 1 and os.environ.get("FLGP_BENCH_E2E_RANK") == "1"     # one-GPU box: the same code with a NULL communicator
    if ((world > 1 and comm is not None) or rehearse_rank_e2e) and not args.no_e2e:
        # N > 1 at the host boundary: every rank hands its rows of X over as host memory and gets its rows of H back
        # (flgp_heat_kernel_covariance_rank: upload, agreement, sharded path, pinned pipelined copy) -- each rank has its
        # own PCIe link, so this is the number that should scale with N.  MAX over ranks, best of 3; never part of `value`.
        try:
            res = None
            torch.cuda.empty_cache()
            U_h = np.asfortranarray(np.column_stack([np.ascontiguousarray(U.t().cpu().numpy()), num_class.cpu().numpy()]))
            X_h = np.asfortranarray(X_np)
            H_h = np.empty((n_loc, args.m), order="F")
            best = None
            for _ in range(3):
                barrier()
                t0 = time.perf_counter()
                rc = L.flgp_heat_kernel_covariance_rank(comm, X_h.ctypes.data, n_loc, n_loc, n, lo, args.m, d, U_h.ctypes.data, s, d + 1,
                                                        args.r, args.t, args.K, b"lae", b"cluster-normalized", 1, 0.1,
                                                        H_h.ctypes.data, n_loc, None)
                barrier()
                tt = torch.tensor([(time.perf_counter() - t0) * 1e3, float(rc != 0)], dtype=torch.float64, device=device)
                if world > 1:
                    torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)      # (every rank learns of a failure: no rank raises alone)
                if tt[1].item() != 0.0:
                    raise RuntimeError("a rank failed: " + (L.flgp_last_error().decode("utf-8", "replace") if rc else "another rank"))
                best = float(tt[0].item()) if best is None else min(best, float(tt[0].item()))
            out["t_e2e_rank_ms" if rehearse_rank_e2e else "t_e2e_ms"] = best
            out["t_e2e_rank_note" if rehearse_rank_e2e else "t_e2e_note"] = ("flgp_heat_kernel_covariance_rank on every rank, host pointers in / out (per rank: X %.0f MB up, H %.2f GB "
                                 "down through the pinned ring), max over ranks, best of 3 calls" % (X_h.nbytes / 1e6, H_h.nbytes / 1e9))
            del H_h
        except Exception as e:  # pragma: no cover
            out["t_e2e_ms"] = None
            out["t_e2e_note"] = "failed: %r" % (e,)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        nc = min(args.cpu_rows, n_loc)
        U_np = np.ascontiguousarray(U.t().cpu().numpy())
        out["cpu_baseline"] = cpu_baseline(args, np.asfortranarray(X_np[:nc]), np.asfortranarray(U_np),
                                           num_class.cpu().numpy())
    if rank == 0:
        if args.verbose:
            print(json.dumps(out, indent=1), file=sys.stderr)
        print(json.dumps(out))
    if comm is not None:
        L.flgp_comm_destroy(comm)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
