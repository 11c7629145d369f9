from microbeseg_amd.utils.utils import *  # noqa: F401,F403
from microbeseg_amd.utils.utils import zero_pad_model_input, min_max_normalization, unique_path, write_train_info  # noqa: F401
