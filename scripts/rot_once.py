"""The solver's rotation (5000 x 256)(256 x 256) a few times -- for rocprofv3 counter passes.  usage: rot_once.py [rows] [knob=value ...]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flgp_amd import _lib
L = _lib.lib()
M = 5000
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("="); L.flgp_set_tuning(k.encode(), int(v))
    else:
        M = int(kv)
st = torch.cuda.current_stream().cuda_stream
b = 256
X = torch.randn(b, M, dtype=torch.float64, device="cuda"); W = torch.randn(b, b, dtype=torch.float64, device="cuda")
O = torch.empty(b, M, dtype=torch.float64, device="cuda")
work = torch.empty(1 << 22, dtype=torch.float64, device="cuda")
for _ in range(20):
    _lib.check(L.flgp_dev_gemm(st, M, b, b, 1.0, X.data_ptr(), 1, M, W.data_ptr(), 1, b, 0.0, None, 0, 0, O.data_ptr(), 1, M, work.data_ptr(), work.numel()))
torch.cuda.synchronize()
print("done")
