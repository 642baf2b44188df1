import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import make_case
from flgp_amd.pipeline import HipStages
from oracle import flgp_oracle as oracle
n, d, s, r, window = 3000, 3, 257, 7, int(sys.argv[1]) if len(sys.argv) > 1 else 100
st = HipStages("cuda:0")
X, U0, U = make_case(n, d, s, r, seed=78)
ei, zn = oracle.cross_similarity(X, U, r, gl="normalized")
av, _ = oracle.scale_A(ei, zn, s)
d_ei = torch.from_numpy(ei).cuda(); d_ev = torch.from_numpy(av).cuda()
csc = st.csc(d_ei, s)
G0 = st.gram(d_ei, d_ev, csc).cpu().numpy()
st.L.flgp_set_tuning(b"sparse_window", window)
G1 = st.gram(d_ei, d_ev, csc).cpu().numpy()
Go = oracle.gram(ei, av, s)
print("unwindowed == oracle:", np.array_equal(G0, Go), " windowed == oracle:", np.array_equal(G1, Go))
bad = np.argwhere(G1 != Go)
print(len(bad), "bad; first:", bad[:12].tolist())
print("bad by (tensor dim0 // window):", np.bincount(bad[:, 0] // window), " by (dim1 // window):", np.bincount(bad[:, 1] // window))
cnt = np.bincount(ei.ravel(), minlength=s)
print("entries per column of some bad dim0:", [int(cnt[i]) for i in bad[:8, 0]], "dim1:", [int(cnt[i]) for i in bad[:8, 1]])
