"""Experiment: cost of block-Jacobi round launches with and without rotations (b = 256)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flgp_amd import _lib
from flgp_amd.pipeline import HipStages
S = HipStages("cuda:0"); L = S.L
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    L.flgp_set_tuning(k.encode(), int(v))

def q(name):
    c = ctypes.c_int(0); ms = ctypes.c_double(0); w = ctypes.c_double(0)
    L.flgp_prof_query(name.encode(), ctypes.addressof(c), ctypes.addressof(ms), ctypes.addressof(w))
    return c.value, ms.value

for b in (256,):
    for kind in ("diagonal", "random"):
        rng = np.random.default_rng(0)
        if kind == "diagonal":
            G = np.diag(np.linspace(1.0, 0.1, b))
        else:
            A = rng.normal(size=(b, b)); G = A @ A.T / b
        dG = torch.from_numpy(G).cuda()
        try:
            S.eig_topk(dG, b)
        except Exception:
            pass
        L.flgp_prof_reset(); L.flgp_prof_enable(2)
        try:
            eig, V, info = S.eig_topk(dG, b)
            sw = info['outer_iterations']
        except Exception as e:          # debug knobs that break convergence: 60 sweeps were run
            sw = 60
        torch.cuda.synchronize(); L.flgp_prof_enable(0)
        c, ms = q("jacobi_eig")
        print(f"b={b} {kind:9s}: jacobi_eig {ms:.3f} ms, sweeps={sw}, per sweep {ms/max(sw,1):.3f} ms", flush=True)
