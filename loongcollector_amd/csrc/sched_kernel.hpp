// sched_kernel.hpp -- length-aware line scheduling for ragged batches (included by gpu_runtime.hip only).
//
// The TDFA engine gives every lane one line; a wavefront runs until its longest line is done, so a batch whose
// line lengths vary (BASELINE configs 3 and 5: 128-4096 B) wastes lanes in file order (measured: 0.95 TB/s vs
// 1.83 TB/s for the same lines sorted by length).  These three kernels build a permutation `order` that groups
// lines of similar length -- a counting sort on ceil(len/32), longest first (so the tail of the grid is made of
// short lines) -- and the match kernels read their line index through it.  Cost: 8 B of traffic per line.
#pragma once

#include <hip/hip_runtime.h>

#include <stdint.h>

constexpr int kSchedBlock = 256;
constexpr int kSchedBuckets = 256;  // bucket b holds lengths [32*(255-b) .. 32*(255-b)+31]; b = 0: >= 8160 bytes

__device__ __forceinline__ uint32_t schedBucket(uint32_t len) {
    const uint32_t q = len >> 5;
    return (kSchedBuckets - 1) - (q > uint32_t(kSchedBuckets - 1) ? uint32_t(kSchedBuckets - 1) : q);  // longest first
}
__device__ __forceinline__ uint32_t schedLineLen(const uint32_t* __restrict__ off, const uint32_t* __restrict__ len,
                                                 uint32_t sepBytes, uint32_t i) {
    return len ? len[i] : off[i + 1] - off[i] - sepBytes;
}

// hist[0..256) must be zero on entry
__global__ __launch_bounds__(kSchedBlock) void sched_hist_kernel(const uint32_t* __restrict__ off,
                                                                 const uint32_t* __restrict__ len, uint32_t sepBytes,
                                                                 uint32_t n, const uint32_t* __restrict__ nPtr,
                                                                 uint32_t* __restrict__ hist) {
    __shared__ uint32_t local[kSchedBuckets];
    if (nPtr) n = *nPtr < n ? *nPtr : n;
    local[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * kSchedBlock + threadIdx.x; i < n; i += gridDim.x * kSchedBlock)
        atomicAdd(&local[schedBucket(schedLineLen(off, len, sepBytes, i))], 1u);
    __syncthreads();
    if (local[threadIdx.x]) atomicAdd(&hist[threadIdx.x], local[threadIdx.x]);
}

// cursor[b] = exclusive prefix sum of hist; one workgroup of 256
__global__ __launch_bounds__(kSchedBuckets) void sched_scan_kernel(const uint32_t* __restrict__ hist,
                                                                   uint32_t* __restrict__ cursor) {
    __shared__ uint32_t s[kSchedBuckets];
    s[threadIdx.x] = hist[threadIdx.x];
    __syncthreads();
    uint32_t sum = 0;
    for (uint32_t i = 0; i < threadIdx.x; ++i) sum += s[i];
    cursor[threadIdx.x] = sum;
}

// order[slot] = line; every workgroup reserves one contiguous range per bucket (one global atomic per
// (workgroup, bucket)), lanes rank themselves inside it with LDS atomics
__global__ __launch_bounds__(kSchedBlock) void sched_scatter_kernel(const uint32_t* __restrict__ off,
                                                                    const uint32_t* __restrict__ len,
                                                                    uint32_t sepBytes, uint32_t n,
                                                                    const uint32_t* __restrict__ nPtr,
                                                                    uint32_t* __restrict__ cursor,
                                                                    uint32_t* __restrict__ order) {
    __shared__ uint32_t local[kSchedBuckets];
    __shared__ uint32_t base[kSchedBuckets];
    if (nPtr) n = *nPtr < n ? *nPtr : n;
    const uint32_t chunk = gridDim.x * kSchedBlock;
    for (uint32_t start = blockIdx.x * kSchedBlock; start < n; start += chunk) {
        local[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t i = start + threadIdx.x;
        uint32_t b = 0, rank = 0;
        if (i < n) {
            b = schedBucket(schedLineLen(off, len, sepBytes, i));
            rank = atomicAdd(&local[b], 1u);
        }
        __syncthreads();
        if (local[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], local[threadIdx.x]);
        __syncthreads();
        if (i < n) order[base[b] + rank] = i;
        __syncthreads();
    }
}
