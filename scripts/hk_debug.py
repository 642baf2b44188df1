"""GPU box: where does the panel kernel differ from the tiled GEMM?  Prints the pattern of mismatching elements."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from flgp_amd import _lib
from flgp_amd.pipeline import HipStages
S = HipStages(torch.device("cuda", 0)); L = _lib.lib()
n0, n1, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = torch.Generator(device="cuda"); g.manual_seed(1)
V = torch.randn((K, n0), dtype=torch.float64, device="cuda", generator=g)
vals = torch.linspace(1.0, 0.4, K, dtype=torch.float64, device="cuda")
V1 = V[:, :n1].contiguous()
Hp = S.hk(vals, 1.7, V, V1).cpu().numpy()       # (n1, n0): H[b, a]
L.flgp_set_tuning(b"hk_panel", 0)
Hg = S.hk(vals, 1.7, V, V1).cpu().numpy()
bad = Hp != Hg
print("mismatch", bad.sum(), "of", bad.size)
if bad.any():
    b, a = np.nonzero(bad)
    for name, v, mod in (("b % 32", b, 32), ("b // 32 (pair)", b // 32, None), ("a % 64", a, 64), ("a // 64 (panel)", a // 64, None)):
        x = v % mod if mod else v
        h = np.bincount(x)
        print(name, "bad counts:", h[:80])
    i = np.argmax(bad.ravel()); bb, aa = divmod(i, bad.shape[1])
    print("first bad (b, a) =", bb, aa, Hp[bb, aa], Hg[bb, aa])
    # is the panel result a permutation / shift of the right one?
    row = Hp[bb]; 
    for sh in range(-3, 4):
        print("  shift a by", sh, "matches:", int((np.roll(Hg[bb], sh) == row).sum()))
    col = Hp[:, aa]
    print("  is Hp[:, a] found among rows of Hg[:, a]?", [int(np.argmin(np.abs(Hg[:, aa] - c))) for c in col[:40]])
