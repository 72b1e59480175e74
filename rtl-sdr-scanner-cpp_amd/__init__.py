"""MI355X spectral-scan engine: host-side Python plumbing around libspecscan.so (C ABI in
include/specscan.h). The compute path is hand-written HIP for gfx950 under csrc/; Python is used only
for tests, the benchmark and the one-process-per-GPU launcher."""
from . import abi, build, channelizer, dist, engine, replay, synth, tracker  # noqa: F401
from .engine import SpectrumEngine, load_library  # noqa: F401

__all__ = ["abi", "build", "channelizer", "dist", "engine", "replay", "synth", "tracker", "SpectrumEngine", "load_library"]
