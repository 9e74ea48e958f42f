"""blazr_amd -- MI355X-native quantised forward path for blazr (host mirror over libblazr_hip.so).

Only what the hot path needs: `csrc/` (HIP kernels + the C-ABI of include/blazr_hip.h), `runtime` (ctypes mirror of
the boostr surface blazr's engine calls) and `synth` (seeded synthetic checkpoints in the reference loaders' formats).
"""
__all__ = ["runtime", "synth"]
