"""Python plumbing for tests, tools and bench.py (ctypes over the C ABI of lib/liblc_regex_gpu.so).  The product is the library."""
