"""Host-side logic that needs no GPU: initial conditions, state container, sink partitioning."""
import numpy as np
import pytest

import bachelor_thesis_n_body_amd as nb
from bachelor_thesis_n_body_amd.dist import partition


def test_plummer_is_seeded_and_centred():
    a, b, c = nb.plummer_sphere(1000, 3), nb.plummer_sphere(1000, 3), nb.plummer_sphere(1000, 4)
    assert np.array_equal(a.pos, b.pos) and np.array_equal(a.vel, b.vel)
    assert not np.array_equal(a.pos, c.pos)
    assert np.allclose(a.mass, 1e-3) and a.mass.sum() == pytest.approx(1.0)
    assert np.abs(a.pos.mean(0)).max() < 1e-14 and np.abs(a.vel.mean(0)).max() < 1e-14


def test_plummer_radial_profile():
    s = nb.plummer_sphere(20000, 5)
    r = np.linalg.norm(s.pos, axis=1)
    # half-mass radius of a Plummer model: a / sqrt(2^{2/3} - 1) = 1.3048 a (truncation at 10a shifts it < 1 %)
    assert np.median(r) == pytest.approx(1.3048 * nb.PLUMMER_A, rel=0.03)
    assert r.max() <= 10.5 * nb.PLUMMER_A
    # every particle is bound: v < v_esc
    v = np.linalg.norm(s.vel + 0.0, axis=1)
    vesc = np.sqrt(2.0) * (r * r + nb.PLUMMER_A ** 2) ** -0.25
    assert np.all(v < 1.05 * vesc)


def test_plummer_dtype_cast():
    s = nb.plummer_sphere(64, 0, np.float32)
    assert s.pos.dtype == np.float32 and s.vel.dtype == np.float32 and s.mass.dtype == np.float32
    d = nb.plummer_sphere(64, 0)
    assert np.array_equal(s.pos, d.pos.astype(np.float32))
    with pytest.raises(ValueError):
        nb.plummer_sphere(0)


def test_state_copy_is_deep():
    s = nb.plummer_sphere(8, 0)
    c = s.copy()
    c.pos += 1.0
    assert not np.array_equal(s.pos, c.pos) and s.n == 8
    assert s.astype(np.float32).pos.dtype == np.float32


@pytest.mark.parametrize("n,world", [(1048576, 8), (10, 4), (7, 8), (1, 1), (1000, 3), (0, 2)])
def test_partition_is_contiguous_and_balanced(n, world):
    counts, offsets = partition(n, world)
    assert sum(counts) == n and len(counts) == world
    assert max(counts) - min(counts) <= 1
    assert offsets[0] == 0
    for r in range(1, world):
        assert offsets[r] == offsets[r - 1] + counts[r - 1]
