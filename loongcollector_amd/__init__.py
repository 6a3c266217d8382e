"""Python plumbing for tests, tools and bench.py (ctypes over the C ABI of lib/liblc_regex_gpu.so).  The product is the library."""
import os

# The HIP runtime reads GPU_MAX_HW_QUEUES (default 4) when it initialises; the library sets it to 16 when it is loaded
# (csrc/gpu_runtime.hip), which is too late in a process whose torch has already touched the GPU.  Same default here, first.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
