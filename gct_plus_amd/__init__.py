"""gct_plus_amd -- MI355X-native (gfx950) Transformer-VAE training-step hot path of
GCT-Plus behind the reference's Python entry points.

Importing the package is cheap and GPU-free; the HIP library
(`gct_plus_amd/libgctplus_hip.so`, C ABI declared in `include/gctplus_hip.h`) is
loaded on first use by `gct_plus_amd._lib` and its absence is a hard error -- there
is no CPU or PyTorch-eager fallback in this package.
"""
__version__ = "0.1.0"
