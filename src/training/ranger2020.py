from microbeseg_amd.training.ranger2020 import Ranger, centralized_gradient  # noqa: F401
