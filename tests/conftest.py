"""pytest configuration: the `gpu` marker separates the parity tests proper (they call the HIP
path through the C ABI on a real MI355X) from everything that runs on CPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import flgp_oracle as O
    O.build()
    return O


def make_case(n, d, s, r, seed, components=5, with_sizes=True):
    """Seeded Gaussian-mixture cloud + random-row anchors (+ 1-NN cluster sizes from the oracle)."""
    from flgp_amd import synth
    from oracle import flgp_oracle as O
    X = synth.gaussian_mixture(n, d, components=components, seed=seed)
    rows = synth.random_anchor_rows(n, s, seed=seed)
    U0 = synth.anchors_from_rows(X, rows)
    if not with_sizes:
        return X, U0, None
    sizes = np.bincount(O.knn(X, U0, 1)[:, 0], minlength=s).astype(float)
    return X, U0, np.asfortranarray(np.hstack([U0, sizes[:, None]]))
