"""Experiment: time the k-NN kernel variants on the GPU and check them against the oracle."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flgp_amd import synth  # noqa: E402
from oracle import flgp_oracle as O  # noqa: E402

L = ctypes.CDLL(os.path.join(ROOT, "flgp_amd", "libflgp_hip.so"))
L.flgp_last_error.restype = ctypes.c_char_p
P = ctypes.c_void_p
L.flgp_dev_anchor_prep.argtypes = [P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P]
L.flgp_dev_knn.argtypes = [P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P, ctypes.c_int, ctypes.c_int, P, P, ctypes.c_int]
L.flgp_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    d, s, r = 16, 5000, 10
    ncheck = 20000
    X = synth.gaussian_mixture(n, d)
    rows = synth.random_anchor_rows(n, s)
    U = synth.anchors_from_rows(X, rows)
    dev = torch.device("cuda:0")
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)  # (d, n) contiguous == column-major n x d
    dU = torch.from_numpy(np.ascontiguousarray(U.T)).to(dev)
    dpad = 16
    dUt = torch.empty((5120, dpad), dtype=torch.float64, device=dev)
    duu = torch.empty(5120, dtype=torch.float64, device=dev)
    didx = torch.empty((r, n), dtype=torch.int32, device=dev)
    ddist = torch.empty((r, n), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    rc = L.flgp_dev_anchor_prep(st, dU.data_ptr(), s, s, d, dUt.data_ptr(), duu.data_ptr())
    assert rc == 0, L.flgp_last_error()
    t0 = time.time()
    oidx, odist = O.knn(X[:ncheck], U, r, output=True)
    print(f"oracle knn on {ncheck} rows: {time.time()-t0:.2f}s ({O.threads()} threads)", flush=True)
    for variant, name in [(6, "P1 A2 KS8"), (7, "P1 A2 KS4"), (8, "P1 A2 KS12"), (9, "P1 A1 KS8"), (10, "P1 A2 KS0")]:
        L.flgp_set_tuning(b"knn_variant", variant)
        didx.zero_(); ddist.zero_()
        rc = L.flgp_dev_knn(st, dX.data_ptr(), n, n, d, dUt.data_ptr(), duu.data_ptr(), s, r, didx.data_ptr(), ddist.data_ptr(), n)
        assert rc == 0, L.flgp_last_error()
        torch.cuda.synchronize()
        gi = didx[:, :ncheck].T.cpu().numpy(); gd = ddist[:, :ncheck].T.cpu().numpy()
        ok_i = np.array_equal(gi, oidx); ok_d = np.array_equal(gd, odist)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        iters = 5
        e0.record()
        for _ in range(iters):
            L.flgp_dev_knn(st, dX.data_ptr(), n, n, d, dUt.data_ptr(), duu.data_ptr(), s, r, didx.data_ptr(), ddist.data_ptr(), n)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        tf = 2.0 * n * s * d / (ms * 1e-3) / 1e12
        print(f"variant {variant} ({name}): {ms:.3f} ms  {tf:.2f} TFLOP/s(f64 dist only)  idx_exact={ok_i} dist_exact={ok_d}", flush=True)


if __name__ == "__main__":
    main()
