"""cp2_amd -- MI355X-native kernels and host code for the CP2 pre-training hot path.

The package is deliberately small: `csrc/` holds the hand-written gfx950 HIP
kernels behind the C ABI declared in include/cp2hip.h, `ops.py` binds them with
ctypes, and `builder.py` / `main.py` mirror the reference's operator surface
(kimathikaai/CP2 builder.py / main.py) for the contrastive path.

There is no CPU fallback: every op raises if libcp2hip.so is missing or a tensor
is not on the GPU.
"""
__version__ = "0.1.0"
