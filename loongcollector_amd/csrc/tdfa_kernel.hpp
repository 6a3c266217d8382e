// tdfa_kernel.hpp -- gfx950 kernel of the tagged-DFA engine (included by gpu_runtime.hip only).
//
// One log line per lane, 64 lines per wavefront, tables in LDS.  A line is walked in aligned 16-byte chunks;
// every chunk is stepped in three phases so that the only serial dependency is the DFA state itself:
//
//   phase 0  class lookups   col[j] = cmap8[byte j]                       16 x (v_add_sdwa + ds_read_u8), independent
//   phase 1  state chain     t = lds32[(t & 0xFFFF) + col[j]]             16 x (v_add_sdwa + ds_read_b32), dependent;
//                            the low half of a transition entry IS the LDS address of the next state's row
//   phase 2  capture writes  regs[(t_j >> 16)][lane] = pos_j              16 x (v_add_sdwa + v_add + ds_write_b32), independent;
//                            the high half of an entry IS the LDS offset of the offset register the transition
//                            stamps; transitions that stamp nothing point at a dummy register, so phase 2 has no
//                            branches and no divergence (lines hit group boundaries at different bytes: a
//                            branch per byte would be taken by some lane almost every byte)
//
// Transitions whose register program is more than "one register = pos" (rare for log regexes) carry a list id
// and a flag instead; a wave that saw such a flag anywhere in the chunk replays phase 2 in order, with branches.
// Capture-offset registers live in LDS as regs[reg][lane]: the register number is data dependent (VGPRs would
// need scratch indexing) and [reg][lane] is bank-conflict free for both phases.
//
// Bytes outside the lane's line (head of the first aligned chunk, tail of the last) take the row's identity
// column.  When every lane of the wavefront is in the middle of its line a wave-uniform branch skips the
// validity selects.  Lines are read with aligned 16-byte global loads through a ring of four chunk registers, so
// three chunks are always ahead of the one being stepped.  An aligned 16-byte chunk holding at least one byte of the line never
// leaves the line's pages, so no load can fault; chunks wholly outside the line are not loaded.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/lc_regex_gpu.h"
#include "device_tables.h"

#ifndef LC_TDFA_MIN_WAVES
#define LC_TDFA_MIN_WAVES 6  // waves per SIMD the register allocator must leave room for (VGPR budget 80)
#endif

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t __attribute__((address_space(3))) * LdsWordPtr;
typedef const uint8_t __attribute__((address_space(3))) * LdsBytePtr;

// general register program (a list of moves); rare for log regexes, kept out of line so the byte loop stays small
template <int BLOCK>
__device__ __forceinline__ void tdfaRunMoveList(uint8_t* smem, uint32_t regsBase, uint32_t list, uint32_t pos,
                                             uint32_t tid) {
    uint32_t* regs = reinterpret_cast<uint32_t*>(smem + regsBase);
    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(smem);
    const uint32_t* opsStart = reinterpret_cast<const uint32_t*>(smem + hdr[TD_OFF_OPSSTART]);
    const uint16_t* ops = reinterpret_cast<const uint16_t*>(smem + hdr[TD_OFF_OPS]);
    const uint32_t at = opsStart[list];
    const uint32_t cnt = ops[at];
#pragma unroll 1
    for (uint32_t i = 0; i < cnt; ++i) {
        const uint32_t w = ops[at + 1 + i];
        const uint32_t dst = w & 0xFF, src = w >> 8;
        const uint32_t val = (src == TD_REG_POS) ? pos : regs[src * BLOCK + tid];
        regs[dst * BLOCK + tid] = val;
    }
}

// a + (t & 0xFFFF)   -- one VALU: SDWA selects the low word of t
__device__ __forceinline__ uint32_t addLowHalf(uint32_t a, uint32_t t) {
    uint32_t r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
        : "=v"(r)
        : "v"(a), "v"(t));
    return r;
}
// a + (t >> 16)
__device__ __forceinline__ uint32_t addHighHalf(uint32_t a, uint32_t t) {
    uint32_t r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
        : "=v"(r)
        : "v"(a), "v"(t));
    return r;
}

// In-order replay of one chunk for wavefronts that met a general register program in it: re-walks the 16 bytes
// from the chunk's entry state and applies every register program at its own byte.  Out of line and rolled up:
// it is rare, and keeping it small keeps the hot loop's register footprint small.
template <int BLOCK>
__device__ __forceinline__ void tdfaReplayChunk(uint8_t* smem, u32x4 q, uint32_t t, uint32_t base, uint32_t L,
                                             uint32_t idCol, uint32_t regsBase, uint32_t tid) {
    const LdsBytePtr cmap = reinterpret_cast<LdsBytePtr>(TD_CMAP_OFFSET);
    const uint32_t regAddr0 = regsBase + tid * 4;
#pragma unroll 1
    for (uint32_t j = 0; j < 16; ++j) {
        const uint32_t word = (j < 8) ? ((j < 4) ? q.x : q.y) : ((j < 12) ? q.z : q.w);
        const uint32_t b = (word >> ((j & 3) * 8)) & 0xFFu;
        const uint32_t col = (base + j < L) ? uint32_t(cmap[b]) : idCol;
        t = *reinterpret_cast<LdsWordPtr>(addLowHalf(col, t));
        if (t & (TD_OP_GENERAL << 16)) tdfaRunMoveList<BLOCK>(smem, regsBase, t >> 17, base + j, tid);
        else *reinterpret_cast<LdsWordPtr>(addHighHalf(regAddr0, t)) = base + j;
    }
}

// steps one aligned 16-byte chunk; CHECKED=false is the mid-line fast path (all 16 bytes belong to the line)
template <int BLOCK, bool CHECKED>
__device__ __forceinline__ uint32_t tdfaStepChunk(uint8_t* smem, const u32x4& q, uint32_t t, uint32_t base, uint32_t L,
                                                  uint32_t idCol, uint32_t regsBase, uint32_t tid) {
    // the blob sits at LDS address 0, so table offsets are LDS addresses
    const LdsBytePtr cmap = reinterpret_cast<LdsBytePtr>(TD_CMAP_OFFSET);
    const uint32_t regAddr0 = regsBase + tid * 4;  // LDS address of regs[0][lane]
    const uint32_t words[4] = {q.x, q.y, q.z, q.w};
    const uint32_t entry = t;
    uint32_t col[16], tt[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {  // phase 0
        const uint32_t b = (words[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
        const uint32_t c4 = cmap[b];
        col[j] = CHECKED ? ((base + j < L) ? c4 : idCol) : c4;
    }
    uint32_t seen = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {  // phase 1
        t = *reinterpret_cast<LdsWordPtr>(addLowHalf(col[j], t));
        tt[j] = t;
        seen |= t;
    }
    if (!__any((seen & (TD_OP_GENERAL << 16)) != 0)) {
#pragma unroll
        for (int j = 0; j < 16; ++j)  // phase 2
            *reinterpret_cast<LdsWordPtr>(addHighHalf(regAddr0, tt[j])) = base + j;
    } else {
        tdfaReplayChunk<BLOCK>(smem, q, entry, base, L, idCol, regsBase, tid);
    }
    return t;
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK, LC_TDFA_MIN_WAVES) void tdfa_match_kernel(const uint8_t* __restrict__ data,
                                                           const uint32_t* __restrict__ off,
                                                           const uint32_t* __restrict__ len, uint32_t sepBytes,
                                                           uint32_t nLines, const uint32_t* __restrict__ blob,
                                                           uint32_t blobBytes, uint32_t nGroupsOut,
                                                           int32_t* __restrict__ caps, uint8_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    {  // stage the tables: 16-byte coalesced copies
        const uint4* src = reinterpret_cast<const uint4*>(blob);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = tid; i < blobBytes / 16; i += BLOCK) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(smem);
    const uint32_t nSlots = hdr[TD_NSLOTS];
    const uint32_t rowBytes = hdr[TD_ROW_BYTES];
    const uint32_t idCol = hdr[TD_ID_COL];
    const uint32_t regsBase = blobBytes;
    uint32_t t = hdr[TD_START_ROW];  // low 16 bits: LDS address of the current state's row

    const uint32_t line = blockIdx.x * BLOCK + tid;
    const bool live = line < nLines;
    uint32_t o = 0, L = 0;
    if (live) {
        o = off[line];
        L = len ? len[line] : off[line + 1] - o - sepBytes;
    }

    const uintptr_t addr = reinterpret_cast<uintptr_t>(data) + o;
    const uint32_t head = uint32_t(addr & 15);
    // address_space(1): the loads must be global_load (vmcnt only).  A generic pointer makes them flat_load, which
    // also counts on lgkmcnt -- every LDS wait of the byte loop would then drain the prefetch.
    typedef const u32x4 __attribute__((address_space(1))) * GlobalChunk;
    const GlobalChunk chunk = reinterpret_cast<GlobalChunk>(addr - head);
    const uint32_t nChunks = L ? (head + L + 15) / 16 : 0;
    // ring of four chunks: while chunk k is stepped, chunks k+1..k+3 are in registers or in flight
    u32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    if (0 < nChunks) c0 = chunk[0];
    if (1 < nChunks) c1 = chunk[1];
    if (2 < nChunks) c2 = chunk[2];
    if (3 < nChunks) c3 = chunk[3];

    for (uint32_t k = 0; k < nChunks; ++k) {
        u32x4 incoming = {0, 0, 0, 0};
        if (k + 4 < nChunks) incoming = chunk[k + 4];
        const uint32_t base = k * 16 - head;  // line offset of byte 0 of the chunk (wraps in the head)
        const bool full = base < L && L - base >= 16;
        if (__all(full)) {
            t = tdfaStepChunk<BLOCK, false>(smem, c0, t, base, L, idCol, regsBase, tid);
        } else {
            t = tdfaStepChunk<BLOCK, true>(smem, c0, t, base, L, idCol, regsBase, tid);
        }
        c0 = c1;
        c1 = c2;
        c2 = c3;
        c3 = incoming;
        if ((t & 0xFFFFu) == TD_TRANS_OFFSET) break;  // dead state (row 0): regex_match can no longer succeed
    }

    if (!live) return;
    const uint16_t* finalId = reinterpret_cast<const uint16_t*>(smem + hdr[TD_OFF_FINALID]);
    const uint8_t* finalMap = smem + hdr[TD_OFF_FINALMAP];
    const uint32_t* regs = reinterpret_cast<const uint32_t*>(smem + regsBase);
    const uint32_t state = ((t & 0xFFFFu) - TD_TRANS_OFFSET) / rowBytes;
    const uint32_t fid = finalId[state];
    const bool matched = (state != 0) && (fid != 0xFFFFu);
    int32_t* out = caps + size_t(line) * 2 * nGroupsOut;
    for (uint32_t s = 0; s < 2 * nGroupsOut; ++s) {
        int32_t val = -1;
        if (matched && s < nSlots) {
            const uint32_t m = finalMap[fid * nSlots + s];
            if (m == TD_REG_POS) val = int32_t(L);
            else if (m != TD_REG_NONE) val = int32_t(regs[m * BLOCK + tid]);
        }
        out[s] = val;
    }
    status[line] = matched ? LC_MATCH : LC_NOMATCH;
}
