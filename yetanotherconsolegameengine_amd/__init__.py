"""yetanotherconsolegameengine_amd — MI355X-native ray-trace core for YetAnotherConsoleGameEngine.

Only what the hot path needs lives here:
  csrc/      host C++ + hand-written gfx950 HIP kernels behind the C-ABI of include/ycge.h
  abi.py     ctypes mirror of that ABI
  scene.py   host-side mirror of the reference's Scene / primitive builder API
  scenes.py  the five BASELINE.json configuration scenes
  renderer.py  RaytraceRenderer mirror (SetCamera / SetFov / Resize / TryFlipAndBlit)
"""
from . import abi  # noqa: F401

__all__ = ["abi"]
