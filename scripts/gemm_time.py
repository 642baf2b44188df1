"""Time flgp_dev_gemm on the eigensolver's shapes as a function of the k depth (fixed cost vs per-stage cost).
usage: python scripts/gemm_time.py [knob=value ...]"""
import sys, torch
sys.path.insert(0, ".")
from flgp_amd import _lib
L = _lib.lib()
for kv in sys.argv[1:]:
    k, v = kv.split("="); L.flgp_set_tuning(k.encode(), int(v))
st = torch.cuda.current_stream().cuda_stream
s, b = 5000, 256
work = torch.empty(64 * 1024 * 1024 // 8 * 4, dtype=torch.float64, device="cuda")
def run(name, M, N, Kd, A, a_s, B, b_s, C, c_s, use_work):
    args = (st, M, N, Kd, 1.0, A.data_ptr(), a_s[0], a_s[1], B.data_ptr(), b_s[0], b_s[1], 0.0, None, 0, 0, C.data_ptr(), c_s[0], c_s[1],
            work.data_ptr() if use_work else None, work.numel() if use_work else 0)
    for _ in range(3): _lib.check(L.flgp_dev_gemm(*args))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); 
    for _ in range(50): L.flgp_dev_gemm(*args)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} M={M} N={N} K={Kd}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
X = torch.randn(b, s, dtype=torch.float64, device="cuda")       # column-major s x b
W = torch.randn(b, b, dtype=torch.float64, device="cuda")
O = torch.empty(b, s, dtype=torch.float64, device="cuda")
T = torch.empty(b, b, dtype=torch.float64, device="cuda")
for Kd in (16, 64, 256):
    run("rotate  (s x b)(b x b)", s, b, Kd, X, (1, s), W, (1, b), O, (1, s), True)
for Kd in (16, 320, 1280, 5000):
    run("gram    (b x s)(s x b)", b, b, Kd, X, (s, 1), X, (1, s), T, (1, b), True)
# per-stage cost of the rotation as a function of the number of row tiles (latency- or bandwidth-bound?)
for M in (640, 1280, 2560, 5000, 10000, 20000):
    Xm = torch.randn(b, M, dtype=torch.float64, device="cuda"); Om = torch.empty(b, M, dtype=torch.float64, device="cuda")
    for Kd in (64, 256):
        run("rotate rows=%d" % M, M, b, Kd, Xm, (1, M), W, (1, b), Om, (1, M), True)
# channel camping?  W with a padded leading dimension (2112 B / 2176 B between its columns instead of 2048 B)
for ldw in (256, 264, 272, 288):
    Wp = torch.randn(b, ldw, dtype=torch.float64, device="cuda")
    run("rotate ldW=%d" % ldw, s, b, 256, X, (1, s), Wp, (1, ldw), O, (1, s), True)
# and X with a padded leading dimension
for lds_ in (5000, 5008, 5024, 5120):
    Xp = torch.randn(b, lds_, dtype=torch.float64, device="cuda"); Op = torch.empty(b, lds_, dtype=torch.float64, device="cuda")
    run("rotate ldX=%d" % lds_, s, b, 256, Xp, (1, lds_), W, (1, b), Op, (1, lds_), True)
# W row-major (k contiguous becomes row contiguous for the kernel: the RC mapping instead of KC)
Wt = torch.randn(b, b, dtype=torch.float64, device="cuda")
run("rotate W row-major", s, b, 256, X, (1, s), Wt, (b, 1), O, (1, s), True)
