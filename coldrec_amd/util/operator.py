"""Name-only counterpart of the reference's util/operator.py, which is dead code there (scalar
similarity helpers imported nowhere -- SURVEY.md F4).  The operator API model files actually use
lives in ``coldrec_amd.util.utils`` and ``coldrec_amd.util.databuilder``."""
from .utils import bpr_loss, l2_reg_loss, next_batch_pairwise, set_seed  # noqa: F401
