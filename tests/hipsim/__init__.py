"""TEST INFRASTRUCTURE: host SIMT interpreter for the library's HIP kernels (see include/hip/hip_runtime.h)."""
