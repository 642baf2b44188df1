"""Experiment: the H contraction's GEMM at several K (stages per tile) and m: where does the idle third of the matrix pipes
come from -- per tile (prologue / epilogue) or per stage?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flgp_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
def run(n, m, K, iters=5):
    A = torch.randn(n * K, dtype=torch.float64, device="cuda"); B = torch.randn(K * m, dtype=torch.float64, device="cuda")
    C = torch.empty(n * m, dtype=torch.float64, device="cuda")
    def call():
        _lib.check(L.flgp_dev_gemm(st, n, m, K, 1.0, A.data_ptr(), 1, n, B.data_ptr(), m, 1, 0.0, None, 0, 0, C.data_ptr(), 1, n, None, 0))
    call(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"n={n} m={m} K={K}: {ms:8.3f} ms  {2.0*n*m*K/ms/1e9:6.2f} TF  ({(K+15)//16} stages per tile)", flush=True)
for K in (64, 200, 208, 400, 800, 1600):
    run(250000 if K > 400 else 1000000, 1024, K)
run(1000000, 1000, 200)
