"""vqengine_amd — MI355X-native (gfx950) offscreen forward-PBR / IBL / post path behind VQEngine's
constant-buffer interface. The compute lives in vqengine_amd/lib/libvqhip.so (hand-written HIP, C ABI in
include/vqhip.h); this package is the thin host side: ctypes binding (capi), ABI structs (abi), seeded
synthetic inputs (synth), the engine-side producers (scene) and the row-tiled multi-GPU mode (tiling). The reference-shaped pass objects
are C++: include/vqhip_passes.hpp."""
from . import abi  # noqa: F401

__all__ = ["abi"]
