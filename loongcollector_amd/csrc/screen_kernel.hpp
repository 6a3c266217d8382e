// screen_kernel.hpp -- yes/no DFA over lines, one line per lane, TABLE IN GLOBAL MEMORY (L2-resident).
//
// The screens in front of the NFA engine (Grok: lcGrokMatchDevice) only need "can this value contain a match"; the relaxed
// whole-pattern screens (regex_handle.cpp lcCompileRelaxedScreen, screen_dfa.cpp) have 1 000 - 20 000 states, far beyond the
// 64 KiB LDS window of the capture kernels, but their tables (u16 next-state per (state, byte class): 0.1 - 2 MB) sit in L2.
// One dependent L2 read per byte and lane is slow per line (~0.3 us a byte) and irrelevant in aggregate: every value of the
// batch is in flight at once, and the pass saves the NFA kernel -- 1000x slower per byte -- most of its values.
//
// Blob (u32 words): SC_* header, 256-byte class map, accept flags u8[nStates], table u16[nStates][nClasses].
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "screen_kernel_layout.h"

constexpr int kScreenBlock = 256;

// in == nullptr: values 0..nIn-1.  Accepted values are appended to out[counters[0]++] (order unspecified).
__global__ __launch_bounds__(kScreenBlock) void dfa_screen_kernel(const uint8_t* __restrict__ data, const uint32_t* __restrict__ off,
                                                                  const uint32_t* __restrict__ len, const uint32_t* __restrict__ in,
                                                                  uint32_t nIn, const uint32_t* __restrict__ blob,
                                                                  uint32_t* __restrict__ out, uint32_t* __restrict__ counters) {
    __shared__ uint8_t cmap[256];
    cmap[threadIdx.x] = reinterpret_cast<const uint8_t*>(blob + SC_HEADER_WORDS)[threadIdx.x];
    __syncthreads();
    const uint32_t k = blockIdx.x * kScreenBlock + threadIdx.x;
    if (k >= nIn) return;
    const uint32_t line = in ? in[k] : k;
    const uint32_t L = len[line];
    const uint32_t ncls = blob[SC_NCLASSES], sink = blob[SC_SINK];
    uint32_t state = blob[SC_START];
    const uint8_t* accept = reinterpret_cast<const uint8_t*>(blob) + blob[SC_OFF_ACCEPT];
    const uint16_t* table = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(blob) + blob[SC_OFF_TABLE]);
    const uintptr_t addr = reinterpret_cast<uintptr_t>(data) + off[line];
    const uint32_t head = uint32_t(addr & 15);
    const uint4* q = reinterpret_cast<const uint4*>(addr - head);  // (aligned 16-byte reads, like the capture kernels)
    const uint32_t total = L ? head + L : 0;
    for (uint32_t pos = 0; pos < total && state != sink && state != 0; pos += 16) {
        const uint4 v = *q++;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) {
            const uint32_t bi = pos + j;
            if (bi >= head && bi < total) state = table[state * ncls + cmap[(w[j >> 2] >> ((j & 3) * 8)) & 0xFFu]];
        }
    }
    if (state == sink || (state != 0 && accept[state])) out[atomicAdd(&counters[0], 1u)] = line;
}
