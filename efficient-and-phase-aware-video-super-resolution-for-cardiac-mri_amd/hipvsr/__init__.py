"""MI355X-native engine for the RefineNet forward/backward hot path (HIP kernels behind a C ABI)."""
