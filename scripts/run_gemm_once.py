"""Runs the G*Q product (5000 x 256 x 5000, eigensolver layout) a few times -- for rocprofv3 counter passes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flgp_amd import _lib
L = _lib.lib()
sizes = [5120]
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "s":
        sizes = [int(x) for x in v.split(",")]
    else:
        L.flgp_set_tuning(k.encode(), int(v))
st = torch.cuda.current_stream().cuda_stream
b = 256
for s in sizes:
  A = torch.randn(s * s, dtype=torch.float64, device="cuda")
  B = torch.randn(s * b, dtype=torch.float64, device="cuda")
  C = torch.empty(s * b, dtype=torch.float64, device="cuda")
  W = torch.empty(8 * s * b, dtype=torch.float64, device="cuda")
  for _ in range(5):
      _lib.check(L.flgp_dev_gemm(st, s, b, s, 1.0, A.data_ptr(), 1, s, B.data_ptr(), 1, s, 0.0, None, 0, 0, C.data_ptr(), 1, s,
                                 W.data_ptr(), 8 * s * b))
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(10):
      _lib.check(L.flgp_dev_gemm(st, s, b, s, 1.0, A.data_ptr(), 1, s, B.data_ptr(), 1, s, 0.0, None, 0, 0, C.data_ptr(), 1, s,
                                 W.data_ptr(), 8 * s * b))
  e1.record(); torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / 10
  print(f"s={s}: {ms*1e3:.1f} us  {2.0*s*s*b/ms/1e9:.2f} TF")

