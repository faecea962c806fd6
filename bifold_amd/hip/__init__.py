"""HIP boundary of bifold_amd: `lib` binds libbifold_hip.so over ctypes, `ops` wraps it for tensors."""
