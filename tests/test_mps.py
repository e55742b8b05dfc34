"""MPS ingestion (torchpdlp_amd/mps.py) against the reference's own loader output (tests/golden/mps.npz, recorded
from /root/reference/PDLP/util.py:mps_to_standard_form on the files in tests/golden/mps/) and against HiGHS."""
import os

import numpy as np
import pytest
import torch

from tests.conftest import GOLDEN
from torchpdlp_amd.mps import mps_to_standard_form, parse_mps

MPS_DIR = os.path.join(GOLDEN, "mps")


@pytest.mark.parametrize("name", ["afiro.mps", "features.mps", "all_eq.mps", "all_ineq.mps"])
def test_matches_reference_loader(golden, name):
    g = golden("mps.npz").group(name)
    c, K, q, m_ineq, l, u = mps_to_standard_form(os.path.join(MPS_DIR, name), device="cpu")
    assert c.dtype == torch.float32 and c.shape == (g["K"].shape[1], 1) and q.shape == (g["K"].shape[0], 1)
    assert m_ineq == int(g["m_ineq"]) and K.shape == g["K"].shape
    np.testing.assert_array_equal(K.to_dense().numpy(), g["K"])          # same rows, same order, same signs
    np.testing.assert_array_equal(c.numpy().ravel(), g["c"])
    np.testing.assert_array_equal(q.numpy().ravel(), g["q"])
    np.testing.assert_array_equal(l.numpy().ravel(), g["l"])
    np.testing.assert_array_equal(u.numpy().ravel(), g["u"])


def test_afiro_shape_and_highs_objective():
    """BASELINE.json configs[0]: K is 27 x 32 with 19 '>=' rows, 83 non-zeros; optimum -464.753142857 (HiGHS)."""
    c, (rowptr, colidx, vals, m, n), q, m_ineq, l, u = parse_mps(os.path.join(MPS_DIR, "afiro.mps"))
    assert (m, n, m_ineq, len(vals)) == (27, 32, 19, 83)
    from scipy.optimize import linprog
    import scipy.sparse as sp
    K = sp.csr_matrix((vals, colidx, rowptr), shape=(m, n))
    res = linprog(c, A_ub=-K[:m_ineq], b_ub=-q[:m_ineq], A_eq=K[m_ineq:], b_eq=q[m_ineq:],
                  bounds=list(zip(l, [None if np.isinf(v) else v for v in u])), method="highs")
    assert res.status == 0 and abs(res.fun - (-464.7531428571)) < 1e-6


def test_marker_lines(golden):
    g = golden("mps.npz")
    assert str(g["marker.mps/error"]) == "ValueError"                   # what the reference does
    with pytest.raises(ValueError):
        mps_to_standard_form(os.path.join(MPS_DIR, "marker.mps"), device="cpu")
    c, K, q, m_ineq, l, u = mps_to_standard_form(os.path.join(MPS_DIR, "marker.mps"), device="cpu", compat=False)
    assert K.shape == (1, 2) and m_ineq == 1                             # LP relaxation when asked for standard semantics


def test_standard_bound_semantics_differ_only_where_documented():
    a = mps_to_standard_form(os.path.join(MPS_DIR, "features.mps"), device="cpu", compat=True)
    b = mps_to_standard_form(os.path.join(MPS_DIR, "features.mps"), device="cpu", compat=False)
    la, ua, lb, ub = a[4].ravel(), a[5].ravel(), b[4].ravel(), b[5].ravel()
    # columns XA..XE: XD is FR, XE is MI
    assert la.tolist() == [-1.0, 0.0, 2.0, 0.0, 0.0] and lb.tolist()[:3] == [-1.0, 0.0, 2.0]
    assert lb[3] == -np.inf and lb[4] == -np.inf and torch.equal(ua, ub)
    assert torch.equal(a[1].to_dense(), b[1].to_dense())
