"""microbeseg_amd — MI355X-native (gfx950) implementation of microbeSEG's U-Net train / infer / watershed hot path.

Layout: ``csrc/`` hand-written HIP kernels + C ABI (``libmseg_hip.so``), ``engine.py`` kernel sequencing,
``utils/``, ``training/``, ``inference/`` = host-side mirror of the reference's ``src/*`` API for this path.
"""
__version__ = "0.1.0"
