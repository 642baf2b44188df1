"""Time the solver's rotation on the general GEMM and on csrc/rot.hip.  usage: rot_time.py [s]"""
import sys, torch
sys.path.insert(0, ".")
from flgp_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
s = 5000
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("="); L.flgp_set_tuning(k.encode(), int(v))
    else:
        s = int(a)
b = 256
X = torch.randn(b, s, dtype=torch.float64, device="cuda"); X2 = torch.randn(b, s, dtype=torch.float64, device="cuda")
W = torch.randn(b, b, dtype=torch.float64, device="cuda"); WT = W.t().contiguous()
O = torch.empty(b, s, dtype=torch.float64, device="cuda"); O2 = torch.empty(b, s, dtype=torch.float64, device="cuda")
work = torch.empty(1 << 22, dtype=torch.float64, device="cuda")
def timeit(name, fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:36s} {e0.elapsed_time(e1) / 100 * 1e3:7.1f} us")
timeit("gemm   X W", lambda: L.flgp_dev_gemm(st, s, b, b, 1.0, X.data_ptr(), 1, s, W.data_ptr(), 1, b, 0.0, None, 0, 0, O.data_ptr(), 1, s, work.data_ptr(), work.numel()))
timeit("rotate X W", lambda: L.flgp_dev_rotate(st, s, b, 1.0, X.data_ptr(), None, WT.data_ptr(), 0.0, None, O.data_ptr(), None))
timeit("gemm pair", lambda: L.flgp_dev_gemm_pair(st, s, b, b, 1.0, X.data_ptr(), X2.data_ptr(), 1, s, W.data_ptr(), W.data_ptr(), 1, b, O.data_ptr(), O2.data_ptr(), 1, s))
timeit("rotate pair", lambda: L.flgp_dev_rotate(st, s, b, 1.0, X.data_ptr(), X2.data_ptr(), WT.data_ptr(), 0.0, None, O.data_ptr(), O2.data_ptr()))
timeit("gemm   E - X W (in place)", lambda: L.flgp_dev_gemm(st, s, b, b, -1.0, X.data_ptr(), 1, s, W.data_ptr(), 1, b, 1.0, O.data_ptr(), 1, s, O.data_ptr(), 1, s, work.data_ptr(), work.numel()))
timeit("rotate E - X W (in place)", lambda: L.flgp_dev_rotate(st, s, b, -1.0, X.data_ptr(), None, WT.data_ptr(), 1.0, O.data_ptr(), O.data_ptr(), None))
T = torch.empty(b, b, dtype=torch.float64, device="cuda")
gw = torch.empty(64 * b * b, dtype=torch.float64, device="cuda")
timeit("gemm   X^T X2 (split + reduce)", lambda: L.flgp_dev_gemm(st, b, b, s, 1.0, X.data_ptr(), s, 1, X2.data_ptr(), 1, s, 0.0, None, 0, 0, T.data_ptr(), 1, b, gw.data_ptr(), gw.numel()))
timeit("gram_small X^T X2", lambda: L.flgp_dev_gram_small(st, s, b, X.data_ptr(), X2.data_ptr(), T.data_ptr(), gw.data_ptr(), gw.numel()))
