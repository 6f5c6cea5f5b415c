"""mvil-fusion_amd -- MI355X-native sliding-window factor-graph backend (hot path of mVIL-Fusion).

Holds only what the path needs: csrc/ (HIP kernels + the C-ABI library libvilsolve.so), the ctypes
mirror of include/vilsolve.h (abi.py), the loader (lib.py) and the synthetic-window generator
(synth.py).  The directory name has a hyphen, so it is registered as module `mvil_fusion_amd` by
__graft_entry__.load_package().
"""
__all__ = ["abi", "lib", "synth"]
