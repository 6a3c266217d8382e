// pipeline_kernel.hpp -- the step AFTER the parser, on the parser's output, without leaving the device (included by
// gpu_runtime.hip only).
//
// The reference's benchmark pipeline (test/benchmark/local/test_cases/performance_file_to_blackhole_loongcollector/
// loongcollector.yaml:7-27) is  split -> processor_parse_regex_native -> processor_filter_regex_native on a key the parser just
// produced (FilterKey: user_agent, FilterRegex: ^no-agent$).  ProcessorFilterNative::IsMatched (ProcessorFilterNative.cpp:258-286)
// asks regex_match(value of key) for every rule; the value of a parsed key is the span of its capture group, and that span is
// already on the device.  One lane per line: lines the parser matched walk each rule's yes/no DFA (screen_kernel_layout.h: the state
// graph of the rule's tagged DFA, accept = "final at end of input" = regex_match) over the span; survivors are packed as
// [line, offset, length, 2G capture offsets] so that only they travel back.  Byte work on L2-hot data; no LDS tables needed
// beyond the class maps.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/lc_regex_gpu.h"
#include "screen_kernel_layout.h"

constexpr int kSpanFilterBlock = 256;
constexpr uint32_t kSpanFilterMax = 8;

struct SpanFilterDev {
    const uint32_t* blob;  // yes/no DFA of the rule's regex
    uint32_t group;        // capture group of the parse regex whose span is the rule's value (1-based)
    uint32_t pad;
};
struct SpanFilterArgs {
    SpanFilterDev f[kSpanFilterMax];
    uint32_t n;
};

// counts: [0] lines, [1] survivors, [2] lines the parser did not match, [3] lines left undecided (LC_OVERFLOW / LC_GAVE_UP)
__global__ __launch_bounds__(kSpanFilterBlock) void span_filter_pack_kernel(
    const uint8_t* __restrict__ data, const uint32_t* __restrict__ off, uint32_t sepBytes, const uint32_t* __restrict__ nLinesPtr,
    uint32_t maxLines, uint32_t nGroups, const int32_t* __restrict__ caps, const uint8_t* __restrict__ status, SpanFilterArgs args,
    int32_t* __restrict__ packed, uint32_t packedCap, uint32_t* __restrict__ counts) {
    __shared__ uint8_t cmap[kSpanFilterMax][256];
    for (uint32_t f = 0; f < args.n; ++f) cmap[f][threadIdx.x] = reinterpret_cast<const uint8_t*>(args.f[f].blob + SC_HEADER_WORDS)[threadIdx.x];
    __syncthreads();
    uint32_t nLines = *nLinesPtr;
    nLines = nLines < maxLines ? nLines : maxLines;
    const uint32_t i = blockIdx.x * kSpanFilterBlock + threadIdx.x;
    if (i == 0) counts[0] = nLines;
    const bool live = i < nLines;
    const uint8_t st = live ? status[i] : uint8_t(LC_NOMATCH);
    bool pass = live && st == LC_MATCH;
    const uint32_t o = live ? off[i] : 0;
    const uint32_t L = live ? off[i + 1] - o - sepBytes : 0;
    const int32_t* c = caps + size_t(i) * 2 * nGroups;
    for (uint32_t f = 0; f < args.n && pass; ++f) {
        const uint32_t* blob = args.f[f].blob;
        const uint32_t ncls = blob[SC_NCLASSES];
        const uint8_t* accept = reinterpret_cast<const uint8_t*>(blob) + blob[SC_OFF_ACCEPT];
        const uint16_t* table = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(blob) + blob[SC_OFF_TABLE]);
        int32_t b = c[2 * (args.f[f].group - 1)], e = c[2 * (args.f[f].group - 1) + 1];
        if (b < 0) b = e = int32_t(L);  // a group that did not take part: boost's {last, last} -- the empty value at the end
        uint32_t state = blob[SC_START];
        const uint8_t* p = data + o;
        for (int32_t k = b; k < e && state != 0; ++k) state = table[state * ncls + cmap[f][p[k]]];
        pass = state != 0 && accept[state];
    }
    // what the host needs to count: parse failures and undecided lines
    const uint64_t failed = __ballot(live && st == LC_NOMATCH);
    const uint64_t undecided = __ballot(live && (st == LC_OVERFLOW || st == LC_GAVE_UP));
    const uint64_t keep = __ballot(pass);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t at = 0;
    if (lane == 0) {
        if (failed) atomicAdd(&counts[2], uint32_t(__popcll(failed)));
        if (undecided) atomicAdd(&counts[3], uint32_t(__popcll(undecided)));
        if (keep) at = atomicAdd(&counts[1], uint32_t(__popcll(keep)));
    }
    at = __shfl(at, 0, 64);
    if (pass) {
        const uint32_t slot = at + __popcll(keep & ((1ull << lane) - 1ull));
        if (slot < packedCap) {
            int32_t* dst = packed + size_t(slot) * (3 + 2 * nGroups);
            dst[0] = int32_t(i);
            dst[1] = int32_t(o);
            dst[2] = int32_t(L);
            for (uint32_t s = 0; s < 2 * nGroups; ++s) dst[3 + s] = c[s];
        }
    }
}

// A read buffer's way up WITHOUT the copy engine: the lanes read the pinned host block through the PCIe mapping (16 bytes each,
// coalesced) and write it to device memory.  hipMemcpyAsync of every runner thread's buffer goes through the device's SDMA queue,
// where the trips of all threads line up behind each other; a kernel does not.
__global__ __launch_bounds__(256) void pinned_upload_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n16) {
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n16; i += uint64_t(gridDim.x) * 256) dst[i] = src[i];
}
