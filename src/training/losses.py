from microbeseg_amd.training.losses import get_loss, ce_dice  # noqa: F401
