"""MI355X-native Conv-TasNet hot path (gfx950 HIP kernels behind a C ABI, PyTorch-ROCm as plumbing).

Drop-in surfaces of the reference kept here: ``ConvTasNet``, ``cal_loss``, ``overlap_and_add``,
``Solver``, ``separate``.  Import name: ``conv_tasnet_amd``.
"""
from ._lib import lib, CtnError, LIB_PATH  # noqa: F401
from .conv_tasnet import ConvTasNet  # noqa: F401
from .pit_criterion import cal_loss, cal_si_snr_with_pit  # noqa: F401
from .utils import overlap_and_add, remove_pad  # noqa: F401
from .ops import gemm_arith, gemm_arithmetic, set_gemm_arith  # noqa: F401

__version__ = "0.1.0"
