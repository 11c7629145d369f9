from microbeseg_amd.evaluation.stats_utils import get_fast_aji_plus  # noqa: F401
