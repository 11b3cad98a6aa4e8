import numpy as np

import _oracle as O
from epic_amd.synthetic import synthetic_grid


def test_numpy_generator_equals_c_generator():
    for m, seed, dens in (([64, 48], 20240601, 0.05), ([9, 11, 13], 3, 0.2), ([130, 257], 99, 0.0)):
        u_c, l_c = O.oracle_synthetic(m, seed, dens)
        u_p, l_p = synthetic_grid(m, seed, dens, chunk=1000)
        assert np.array_equal(u_c, u_p) and np.array_equal(l_c, l_p)


def test_generator_shape_rules():
    u, lk = synthetic_grid([32, 40], 1, 0.05)
    u, lk = u.reshape(32, 40), lk.reshape(32, 40)
    assert lk[0].all() and lk[-1].all() and lk[:, 0].all() and lk[:, -1].all()
    assert u[16, 20] == 0.0 and lk[16, 20] == 1 and (u == 0).sum() == 1
    frac = lk[1:-1, 1:-1].mean()
    assert 0.01 < frac < 0.12
