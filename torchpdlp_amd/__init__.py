"""torchpdlp_amd -- restarted PDHG (PDLP) for linear programs on AMD Instinct MI355X.

The solver hot path of SimplySnap/torchPDLP (PDHG step, adaptive restarts, primal-weight and
step-size updates, KKT residuals) as hand-written HIP kernels behind the reference's own Python
function surface.  ``import torchpdlp_amd`` needs no GPU; calling a solver function does, and
fails loudly if the HIP library is missing (there is no CPU fallback in this package).
"""
from ._native import PdlpError, load as load_native                      # noqa: F401
from .sparse import CsrPair, csr_transpose                               # noqa: F401
from .engine import Comm, PdlpEngine                                     # noqa: F401
from .solver import (STATUS_KKT_LIMIT, STATUS_SOLVED, STATUS_TIME_LIMIT, check_termination,   # noqa: F401
                     pdlp_algorithm, run_pdlp)
from .ops import (KKT_error, adaptive_one_step_pdhg, compute_residuals_and_duality_gap,       # noqa: F401
                  detect_infeasibility, fixed_one_step_pdhg, primal_weight_update, project_lambda_box,
                  spectral_norm_estimate_torch)
from .precondition import ruiz_precondition                              # noqa: F401
from .synthetic import SyntheticLP, gen_lp                               # noqa: F401
from .mps import mps_to_standard_form, parse_mps                         # noqa: F401
from .api import LPResult, solve_lp                                      # noqa: F401
from .spectral_casting import fishnet, sample_points, spectral_cast      # noqa: F401

__version__ = "0.1.0"
