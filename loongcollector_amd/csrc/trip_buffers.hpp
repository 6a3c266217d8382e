// trip_buffers.hpp -- grow-only pinned / device buffers of a runner thread's device trips (multiline_device.hip,
// processor_filter_gpu.cpp): allocated on the thread's first trip, released by lc_thread_release() or when the thread ends.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstddef>

struct TripBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    void release() {
        if (p) (void)(pinned ? hipHostFree(p) : hipFree(p));
        p = nullptr;
        cap = 0;
    }
    hipError_t ensure(size_t bytes) {
        if (p && cap >= bytes) return hipSuccess;
        release();
        const size_t want = bytes + (bytes >> 2) + 256;
        const hipError_t e = pinned ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
};
