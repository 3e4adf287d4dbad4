"""dreamvla_amd -- MI355X-native (gfx950) implementation of DreamVLA's transformer hot path.

Only what the path needs: `csrc/` (hand-written HIP kernels + the C ABI of include/dvla.h), `_lib` (ctypes
binding), `ops` (autograd wrappers) and the host-side mirror of the reference modules.  The drop-in import
path `models.dreamvla_model.DreamVLA` re-exports from here.
"""
__version__ = "0.1.0"
