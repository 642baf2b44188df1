"""CPU: host-side logic of the package (synthetic generators, sharding arithmetic, CSR plumbing)."""
import os

import numpy as np
import pytest

from flgp_amd import api, synth
from flgp_amd.pipeline import shard_bounds


def test_splitmix_reference_values():
    # SplitMix64 with the canonical seed-0 stream: first outputs of the published generator
    out = synth.splitmix64(np.array([0, 0x9E3779B97F4A7C15, 2 * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))
    assert [hex(int(v)) for v in out] == ["0xe220a8397b1dcdaf", "0x6e789e6aa1b965f4", "0x6c45d188009454f"]


def test_generators_are_deterministic_and_blockwise():
    A = synth.gaussian_mixture(1000, 16)
    B = np.vstack([synth.gaussian_mixture(400, 16, row_offset=0), synth.gaussian_mixture(600, 16, row_offset=400)])
    np.testing.assert_array_equal(A, B)      # a rank can generate just its row block
    assert A.flags.f_contiguous and abs(A.mean()) < 1.0
    u = synth.uniform(1, 2, 10000)
    assert 0 < u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.02
    z = synth.normal(1, 2, 20000)
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1) < 0.03


def test_workload_shapes():
    X, y = synth.swiss_roll(500)
    assert X.shape == (500, 3) and y.shape == (500,)
    np.testing.assert_allclose(X.std(0, ddof=1), 1 / np.sqrt(3), rtol=1e-12)
    X, Y = synth.torus(4800)
    assert X.shape == (4800, 2) and set(np.unique(Y)) == {0.0, 1.0}
    rows = synth.random_anchor_rows(1000, 100)
    assert len(set(rows.tolist())) == 100 and rows.min() >= 0 and rows.max() < 1000


@pytest.mark.parametrize("n,world", [(10, 1), (10, 3), (1000000, 8), (7, 8), (0, 2)])
def test_shard_bounds_partition(n, world):
    cuts = [shard_bounds(n, world, r) for r in range(world)]
    assert cuts[0][0] == 0 and cuts[-1][1] == n
    for (a, b), (c, d) in zip(cuts[:-1], cuts[1:]):
        assert b == c and 0 <= b - a <= n // world + 1


def test_csr_plumbing_rejects_ragged_rows():
    import scipy.sparse as sp
    Z = sp.csr_matrix(np.array([[1.0, 2.0, 0.0], [0.0, 0.0, 3.0]]))
    with pytest.raises(ValueError):
        api._csr_parts(Z)
    Z = sp.csr_matrix((np.array([1.0, 2.0, 3.0, 4.0]), np.array([2, 0, 1, 2]), np.array([0, 2, 4])), shape=(2, 3))
    n, s, r, j, x = api._csr_parts(Z)
    assert (n, s, r) == (2, 3, 2)
    np.testing.assert_array_equal(j, [0, 2, 1, 2])        # sorted within rows, as dgRMatrix
    np.testing.assert_array_equal(x, [2.0, 1.0, 3.0, 4.0])


def test_constant_division_identity(tmp_path):
    """(cumsum - 1)/j in the LAE kernels is three FMA-class operations instead of an IEEE division
    (flgp_amd/csrc/lae_dev.h div_const); the identity is checked here on the host's IEEE arithmetic."""
    import subprocess
    src = os.path.join(os.path.dirname(__file__), "c", "div_const_check.c")
    exe = str(tmp_path / "div_const_check")
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", src, "-o", exe, "-lm"], check=True)
    out = subprocess.run([exe, "1000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "mismatches 0" in out.stdout


def test_bounded_host_wait(tmp_path):
    """The eigensolver's host waits poll an event for at most `eig_spin_us` and then block (csrc/host_wait.h,
    used by csrc/eig.hip): a hung kernel must not pin a spinning thread, errors must come back.  The logic is generic
    over the query / block operations and is exercised here on the CPU."""
    import subprocess
    src = os.path.join(os.path.dirname(__file__), "c", "host_wait_check.cc")
    exe = str(tmp_path / "host_wait_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", src, "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "host_wait ok" in out.stdout, out.stdout
