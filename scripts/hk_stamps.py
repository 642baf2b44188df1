"""GPU box: in-kernel s_memtime stamps of the panel contraction (diagnostic variant 5): per panel and wave, the cycles
spent in panel_put + barrier, in the stage loop, and in the closing barrier."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from flgp_amd import _lib
from flgp_amd.pipeline import HipStages
S = HipStages(torch.device("cuda", 0)); L = _lib.lib()
n0, n1, K = 1_000_000, 1000, 200
g = torch.Generator(device="cuda"); g.manual_seed(1)
V = torch.randn((K, n0), dtype=torch.float64, device="cuda", generator=g)
vals = torch.linspace(1.0, 0.4, K, dtype=torch.float64, device="cuda")
V1 = V[:, :n1].contiguous()
dbg = torch.zeros(4 * 8 * 64 * 4, dtype=torch.int64, device="cuda")
p = dbg.data_ptr()
L.flgp_set_tuning(b"hk_panel_var", 5); L.flgp_set_tuning(b"hk_dbg_lo", p & 0xffffffff if (p & 0xffffffff) < 2**31 else (p & 0xffffffff) - 2**32); L.flgp_set_tuning(b"hk_dbg_hi", p >> 32)
for _ in range(3):
    H = S.hk(vals, 10.0, V, V1); torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(4, 8, 64, 4)
for wg in range(2):
    for w in range(8):
        x = d[wg, w, :61]
        put = x[:, 1] - x[:, 0]; loop = x[:, 2] - x[:, 1]; endb = x[:, 3] - x[:, 2]; gap = x[1:, 0] - x[:-1, 3]
        print(f"wg {wg} wave {w}: put+barrier {np.median(put):.0f}  loop {np.median(loop):.0f}  end barrier {np.median(endb):.0f}  between {np.median(gap):.0f}  (s_memtime ticks, 100 MHz => x10 ns); first panels loop {loop[:4]}; total {(x[60,3]-x[0,0])}")
