"""sph_amd -- MI355X-native engine for the SPH frame step of CPearson109/SPH.

  sph_amd.abi     ctypes binding of libsph_hip.so (include/sph_hip.h)
  sph_amd.host    headless mirror of Assets/SPH.cs / SpawnBox.cs / SphereMover.cs
  sph_amd.scenes  the reference's scenes and the BASELINE configurations
  sph_amd/csrc    hand-written gfx950 HIP kernels + the C ABI
"""
from . import abi, host, scenes  # noqa: F401

__all__ = ["abi", "host", "scenes"]
