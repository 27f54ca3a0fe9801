"""MI355X-native registration hot path (FCGF NN -> MNN/GPF -> RANSAC -> Kabsch) behind the
reference's `FR(...)` operator.  All compute goes through the C-ABI library in csrc/ (hand-written
HIP for gfx950); there is no CPU fallback -- calls raise if `liblidarreg.so` cannot be loaded."""
__version__ = "0.1.0"
