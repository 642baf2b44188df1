"""GPU box: the H = V0 diag(exp(-t(1-values))) V1^T contraction alone on random data (for rocprofv3 / variant timing).
usage: python3 scripts/run_hk.py [n0] [n1] [K] [reps] [key=value tuning ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from flgp_amd import _lib
from flgp_amd.pipeline import HipStages

pos = [a for a in sys.argv[1:] if "=" not in a]
n0 = int(pos[0]) if len(pos) > 0 else 1_000_000
n1 = int(pos[1]) if len(pos) > 1 else 1000
K = int(pos[2]) if len(pos) > 2 else 200
reps = int(pos[3]) if len(pos) > 3 else 10
S = HipStages(torch.device("cuda", 0)); L = _lib.lib()
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("="); L.flgp_set_tuning(k.encode(), int(v))
g = torch.Generator(device="cuda"); g.manual_seed(1)
V = torch.randn((K, n0), dtype=torch.float64, device="cuda", generator=g)
vals = torch.linspace(1.0, 0.4, K, dtype=torch.float64, device="cuda")
V1 = V[:, :n1].contiguous()
H = S.hk(vals, 10.0, V, V1); torch.cuda.synchronize()
trash = None
if os.environ.get("HK_TRASH"):
    # what runs between two contractions in bench.py, roughly: other kernels over other memory (cold TLBs / caches, another clock state)
    trash = torch.empty(int(float(os.environ["HK_TRASH"]) * (1 << 30) // 8), dtype=torch.float64, device="cuda")
    A = torch.randn((4096, 4096), dtype=torch.float64, device="cuda", generator=g)
ts = []
for _ in range(reps):
    if trash is not None:
        trash.add_(1.0)
        if os.environ.get("HK_TRASH_MM"):
            for _q in range(int(os.environ["HK_TRASH_MM"])): B_ = A @ A
        torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); H = S.hk(vals, 10.0, V, V1); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts = np.array(ts)
fl = 2.0 * n0 * n1 * K
print(f"hk n0={n0} n1={n1} K={K}: median {np.median(ts):.3f} ms  min {ts.min():.3f}  -> {fl / np.median(ts) / 1e9:.1f} TFLOP/s  ({' '.join(sys.argv[1:])})")
