from microbeseg_amd.utils.unets import *  # noqa: F401,F403
from microbeseg_amd.utils.unets import build_unet, get_weights, UNet, DUNet, Mish, ConvBlock, ConvPool, TranspConvBlock  # noqa: F401
