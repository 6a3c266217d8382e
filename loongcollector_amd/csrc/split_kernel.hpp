// split_kernel.hpp -- line splitting on the GPU (included by gpu_runtime.hip only).
//
// Fuses the step BEFORE the parse processor into the device pipeline: the reference's
// ProcessorSplitLogStringNative::ProcessEvent / GetNextLine
//   (core/plugin/processor/inner/ProcessorSplitLogStringNative.cpp:101-174)
// walks a <= 512 KB read buffer byte by byte on the CPU and creates one event per SplitChar-delimited line.  Here the
// raw buffer is shipped as is; three small kernels produce the line-offset table the match kernels consume:
//   count   : a workgroup scans a 16 KiB tile as four 4 KiB sub-tiles; in each the 256 lanes load 16 CONSECUTIVE bytes each
//             (one fully coalesced 4 KiB request per load instruction -- until round 2 a lane walked its own 64 bytes and
//             every load instruction touched 64 different 64-byte segments); workgroup total of SplitChar hits
//   scan    : exclusive scan of the workgroup totals (one workgroup)
//   scatter : same loads again; hits are ranked in buffer order (sub-tile, lane, byte) and off[k+1] = p+1 is written for
//             the k-th hit
// Resulting table: off[0] = 0, off[i+1] = start of line i+1, off[nLines] chosen so that
// len[i] = off[i+1] - off[i] - 1 for every line (the sep_bytes = 1 convention of lc_regex_match_device), including
// a last line without terminator.  Same line set as the reference: empty lines are lines, a trailing SplitChar does
// not open a new line, an empty buffer has no lines.  HBM-bound: 2 reads of the buffer + 4 B per line.
#pragma once

#include <hip/hip_runtime.h>

#include <stdint.h>

constexpr int kSplitBlock = 256;
constexpr uint32_t kSplitBytesPerLane = 64;
constexpr uint32_t kSplitBytesPerBlock = kSplitBlock * kSplitBytesPerLane;

typedef uint32_t split_u32x4 __attribute__((ext_vector_type(4)));

// 64-bit mask of SplitChar hits in this lane's four 16-byte pieces: bit 16*q + j = byte j of the piece at
// tileBase + q * (kSplitBlock * 16) + lane * 16; bytes past nBytes never hit
__device__ __forceinline__ uint64_t splitPieceBase(uint64_t tileBase, int q) {
    return tileBase + uint64_t(q) * (kSplitBlock * 16) + uint64_t(threadIdx.x) * 16;
}
__device__ __forceinline__ uint64_t splitLaneMask(const uint8_t* __restrict__ data, uint64_t nBytes, uint64_t tileBase,
                                                  uint32_t splitChar) {
    uint64_t mask = 0;
    const uint32_t pattern = splitChar * 0x01010101u;
    const bool aligned = (reinterpret_cast<uintptr_t>(data) & 15) == 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint64_t at = splitPieceBase(tileBase, q);
        if (at >= nBytes) continue;
        uint32_t w[4] = {0, 0, 0, 0};
        if (at + 16 <= nBytes && aligned) {
            const split_u32x4 v = *reinterpret_cast<const split_u32x4*>(data + at);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {  // unaligned base pointer or tail: byte loads
            for (uint32_t j = 0; j < 16 && at + j < nBytes; ++j) w[j >> 2] |= uint32_t(data[at + j]) << ((j & 3) * 8);
            if (splitChar == 0)  // zero padding of the tail must not look like hits
                for (uint32_t j = 0; j < 16; ++j)
                    if (at + j >= nBytes) w[j >> 2] |= 1u << ((j & 3) * 8);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t x = w[k] ^ pattern;  // zero byte <=> hit
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (((x >> (8 * b)) & 0xFFu) == 0) mask |= uint64_t(1) << (q * 16 + k * 4 + b);
        }
    }
    return mask;
}

__device__ __forceinline__ uint32_t splitBlockReduce(uint32_t v, uint32_t* lds) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t total = 0;
    for (int w = 0; w < kSplitBlock / 64; ++w) total += lds[w];
    return total;
}

__global__ __launch_bounds__(kSplitBlock) void split_count_kernel(const uint8_t* __restrict__ data, uint64_t nBytes,
                                                                  uint32_t splitChar, uint32_t* __restrict__ blockHits) {
    __shared__ uint32_t lds[kSplitBlock / 64];
    const uint64_t tileBase = uint64_t(blockIdx.x) * kSplitBytesPerBlock;
    const uint32_t hits = uint32_t(__popcll(splitLaneMask(data, nBytes, tileBase, splitChar)));
    const uint32_t total = splitBlockReduce(hits, lds);
    if (threadIdx.x == 0) blockHits[blockIdx.x] = total;
}

// exclusive scan of blockHits[0..nBlocks) in place; total -> *nHits.  One workgroup of 1024 lanes.
__global__ __launch_bounds__(1024) void split_scan_kernel(uint32_t* __restrict__ blockHits, uint32_t nBlocks,
                                                          uint32_t* __restrict__ nHits) {
    __shared__ uint32_t waveSums[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < nBlocks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nBlocks ? blockHits[i] : 0;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= uint32_t(d)) incl += up;
        }
        if (lane == 63) waveSums[wave] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t w = 0; w < wave; ++w) before += waveSums[w];
        if (i < nBlocks) blockHits[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *nHits = carry;
}

__global__ __launch_bounds__(kSplitBlock) void split_scatter_kernel(const uint8_t* __restrict__ data, uint64_t nBytes,
                                                                    uint32_t splitChar,
                                                                    const uint32_t* __restrict__ blockBase,
                                                                    const uint32_t* __restrict__ nHits,
                                                                    uint32_t* __restrict__ off, uint32_t offCapacity,
                                                                    uint32_t* __restrict__ nLines) {
    __shared__ uint32_t waveHits[4][kSplitBlock / 64];
    const uint64_t tileBase = uint64_t(blockIdx.x) * kSplitBytesPerBlock;
    const uint64_t mask = splitLaneMask(data, nBytes, tileBase, splitChar);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // hits are ranked in buffer order: sub-tile q, then lane, then byte -- one wave scan per sub-tile
    uint32_t incl[4], hits[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        hits[q] = uint32_t(__popc(uint32_t(mask >> (16 * q)) & 0xFFFFu));
        uint32_t v = hits[q];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(v, d, 64);
            if (lane >= uint32_t(d)) v += up;
        }
        incl[q] = v;
        if (lane == 63) waveHits[q][wave] = v;
    }
    __syncthreads();
    uint32_t before = blockBase[blockIdx.x];  // hits of the whole buffer before this tile
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t rank = before + incl[q] - hits[q];
        for (uint32_t w = 0; w < wave; ++w) rank += waveHits[q][w];
        uint32_t m = uint32_t(mask >> (16 * q)) & 0xFFFFu;
        const uint64_t pieceBase = splitPieceBase(tileBase, q);
        while (m) {
            const uint32_t j = uint32_t(__ffs(int(m))) - 1;
            m &= m - 1;
            const uint64_t nextStart = pieceBase + j + 1;  // the line after hit `rank` starts here
            if (uint64_t(rank) + 1 < offCapacity) off[rank + 1] = uint32_t(nextStart);
            ++rank;
        }
        for (uint32_t w = 0; w < kSplitBlock / 64; ++w) before += waveHits[q][w];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t total = *nHits;
        uint32_t lines = 0;
        if (nBytes) {
            off[0] = 0;
            const bool terminated = data[nBytes - 1] == uint8_t(splitChar);
            lines = total + (terminated ? 0 : 1);
            // off[lines] closes the last line: nBytes when the buffer ends with SplitChar (written by the hit itself),
            // nBytes+1 otherwise so that len = off[i+1]-off[i]-1 also holds for the unterminated tail
            if (!terminated && lines < offCapacity) off[lines] = uint32_t(nBytes + 1);
        }
        *nLines = lines;  // may exceed offCapacity-1: the caller checks (LC_ERR_ARG from the host wrapper)
    }
}
