"""Experiment: time flgp_dev_gemm on the shapes the path uses, in several operand layouts."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flgp_amd import _lib  # noqa: E402

L = _lib.lib()
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    L.flgp_set_tuning(k.encode(), int(v))
st = torch.cuda.current_stream().cuda_stream


def run(name, M, N, Kd, a_s, b_s, c_s, work_elems=0, iters=20):
    A = torch.randn(M * Kd, dtype=torch.float64, device="cuda")
    B = torch.randn(Kd * N, dtype=torch.float64, device="cuda")
    C = torch.empty(M * N, dtype=torch.float64, device="cuda")
    W = torch.empty(max(work_elems, 1), dtype=torch.float64, device="cuda")

    def call():
        _lib.check(L.flgp_dev_gemm(st, M, N, Kd, 1.0, A.data_ptr(), a_s[0], a_s[1], B.data_ptr(), b_s[0], b_s[1], 0.0, None, 0, 0,
                                   C.data_ptr(), c_s[0], c_s[1], W.data_ptr() if work_elems else None, work_elems))
    call(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:55s} {ms*1e3:9.1f} us  {2.0*M*N*Kd/ms/1e9:7.2f} TF", flush=True)


s, b, n, m, K = 5000, 256, 1000000, 1000, 200
ws = 8 * s * b
# Y = G Q : G col-major (symmetric), Q col-major s x b, Y col-major
run("G*Q  Q col-major (k-contig B), Y col-major, splitK", s, b, s, (1, s), (1, s), (1, s), ws)
run("G*Q  Q col-major, Y col-major, no split", s, b, s, (1, s), (1, s), (1, s), 0)
# Q stored transposed: Qt is b x s col-major => B(k,j) = Qt[j + k*b]: j-contig
run("G*Qt Qt b x s (j-contig B), Y col-major, splitK", s, b, s, (1, s), (b, 1), (1, s), ws)
run("G*Qt Qt (j-contig B), Yt out (j-contig C), splitK", s, b, s, (1, s), (b, 1), (b, 1), ws)
run("G*Qt Qt (j-contig B), Yt out, no split", s, b, s, (1, s), (b, 1), (b, 1), 0)
# small ones
run("Q^T Z  (b x b, K = s) splitK", b, b, s, (s, 1), (1, s), (1, b), 128 * b * b)
run("Q W   (s x b x b)", s, b, b, (1, s), (1, b), (1, s), 0)
run("b x b x b", b, b, b, (1, b), (1, b), (1, b), 0)
# heat kernel: H(a,b) = sum_k V0(a,k) Vw(b,k)
run("HK n x m x K (col-major H)", n, m, K, (1, n), (m, 1), (1, n), 0, iters=5)
run("HK with V0 stored k-contiguous (n x K row-major)", n, m, K, (K, 1), (m, 1), (1, n), 0, iters=5)
run("HK with both operands k-contiguous", n, m, K, (K, 1), (1, K), (1, n), 0, iters=5)

print("--- split-K sweep for G*Q")
for ns in (2, 3, 4, 5, 6, 8, 10, 12):
    run(f"G*Q col-major, nsplit<={ns}", s, b, s, (1, s), (1, s), (1, s), ns * s * b)
