"""Short-range table: the library's closed-form builder == the oracle's restatement of the reference's
DFT + Newton-Cotes procedure (forcetree.c:3246-3403, ngravs_core.c:72-184), for every law."""
import numpy as np
import pytest


@pytest.mark.parametrize("wiring,ng,pmgrid", [("newton", 1, 64), ("coloyuk", 2, 64), ("yukawa_offdiag", 2, 128),
                                              ("c4", 2, 512), ("c4", 3, 32)])
def test_library_table_equals_oracle(pkg, have_lib, O, wiring, ng, pmgrid):
    cfg = pkg.make_config(n_gravs=ng, periodic=1, pmgrid=pmgrid, box_size=1234.5, wiring=wiring)
    f_lib, p_lib = pkg.shortrange_table(cfg)
    f_orc, p_orc = O.shortrange_table(cfg)
    scale = np.abs(f_orc).max(axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(f_lib - f_orc) / scale) < 1e-11
    assert np.max(np.abs(p_lib - p_orc)) < 1e-10
    if wiring == "yukawa_offdiag":
        assert np.all(f_lib[0, 0] == 0) and np.all(f_lib[1, 1] == 0)   # `none` on the diagonal
