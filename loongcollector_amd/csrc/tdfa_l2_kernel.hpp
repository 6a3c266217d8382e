// tdfa_l2_kernel.hpp -- the tagged DFA for automata that do not fit the 64 KiB LDS window: tables in global memory (L2).
//
// The LDS kernels (tdfa_stream_kernel.hpp) take automata of up to a few hundred states.  Past that a pattern used to fall to the
// thread-list NFA kernel, which is two orders of magnitude slower per byte (one line per WAVEFRONT, several dependent table
// reads per byte-step).  Most such automata are not large by any other measure -- 1 000 to 30 000 states, 0.2 to 8 MB of
// transitions, which an L2 of 4 MB per XCD and the 256 MB Infinity Cache keep close.  So: one line per lane as before, the state
// walk reads  trans[state][class]  from global memory (one dependent L2 access per byte), register programs are interpreted
// from their lists, the offset registers of the 64 lines of a workgroup sit in LDS.  Per line that is slow (~0.3 us a byte);
// in aggregate every line of the batch is in flight at once (dfa_screen_kernel, the same walk without registers: 184 GB/s).
// Semantics: exactly the logical tables (tdfa.hpp), i.e. what tests/helpers/table_interp.py TdfaInterp walks.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/lc_regex_gpu.h"
#include "device_tables.h"
#include "tdfa_l2_layout.h"

constexpr int kTdfaL2Block = 64;

// order (optional): the lines to look at; resume (optional, indexed by line): offset a search resumes at.
// Dynamic LDS: (nRegs) * 64 * 4 bytes of offset registers.
__global__ __launch_bounds__(kTdfaL2Block) void tdfa_l2_kernel(const uint8_t* __restrict__ data, const uint32_t* __restrict__ off,
                                                               const uint32_t* __restrict__ len, uint32_t sepBytes, uint32_t nLines,
                                                               const uint32_t* __restrict__ nLinesPtr, const uint32_t* __restrict__ order,
                                                               const uint32_t* __restrict__ resume, const uint32_t* __restrict__ blob,
                                                               uint32_t nGroupsOut, int32_t* __restrict__ caps,
                                                               uint8_t* __restrict__ status, uint32_t stageBytes,
                                                               uint32_t* __restrict__ pendingFlag, uint32_t launchSeq) {
    extern __shared__ uint32_t regs[];  // [nRegs][64], then (stageBytes != 0) the register programs: opsStart, ops
    __shared__ uint8_t cmap[256];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 256; i += kTdfaL2Block) cmap[i] = reinterpret_cast<const uint8_t*>(blob + TL_HEADER_WORDS)[i];
    const uint32_t nRegs = blob[TL_NREGS], ncls = blob[TL_NCLASSES], nSlots = blob[TL_NSLOTS];
    for (uint32_t r = 0; r < nRegs; ++r) regs[r * kTdfaL2Block + tid] = 0xFFFFFFFFu;  // unset = -1
    // SMALL batches wait for their longest line, and a byte that carries a register program used to cost three more dependent
    // global reads (opsStart[prog], the list's length, each op) behind the transition's own: with the programs in LDS it is the
    // transition's read plus LDS latency.  (Large batches keep them in global memory: there the LDS is better spent on waves.)
    if (stageBytes) {
        uint32_t* dst = regs + nRegs * kTdfaL2Block;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(blob) + blob[TL_OFF_OPSSTART]);
        for (uint32_t i = tid; i < stageBytes / 4; i += kTdfaL2Block) dst[i] = src[i];
    }
    __syncthreads();
    if (nLinesPtr) nLines = *nLinesPtr < nLines ? *nLinesPtr : nLines;
    const uint32_t slot = blockIdx.x * kTdfaL2Block + tid;
    if (slot >= nLines) return;
    const uint32_t line = order ? order[slot] : slot;
    const uint8_t* base = reinterpret_cast<const uint8_t*>(blob);
    const uint32_t* trans = reinterpret_cast<const uint32_t*>(base + blob[TL_OFF_TRANS]);
    // (staged: opsStart sits at the start of the staged block, ops behind it at the distance it has in the blob)
    const uint8_t* staged = reinterpret_cast<const uint8_t*>(regs + nRegs * kTdfaL2Block);
    const uint32_t* opsStart = stageBytes ? reinterpret_cast<const uint32_t*>(staged)
                                          : reinterpret_cast<const uint32_t*>(base + blob[TL_OFF_OPSSTART]);
    const uint16_t* ops = stageBytes ? reinterpret_cast<const uint16_t*>(staged + (blob[TL_OFF_OPS] - blob[TL_OFF_OPSSTART]))
                                     : reinterpret_cast<const uint16_t*>(base + blob[TL_OFF_OPS]);
    const uint32_t o = off[line];
    const uint32_t L = len ? len[line] : off[line + 1] - o - sepBytes;
    uint32_t state = blob[TL_START], from = 0;
    if (resume) {
        from = resume[line];
        from = from < L ? from : L;
        if (from) state = reinterpret_cast<const uint32_t*>(base + blob[TL_OFF_STARTAFTER])[cmap[data[size_t(o) + from - 1]]];
    }
    const uintptr_t addr = reinterpret_cast<uintptr_t>(data) + o + from;
    const uint32_t head = uint32_t(addr & 15);
    const uint4* q = reinterpret_cast<const uint4*>(addr - head);
    const uint32_t total = L > from ? head + (L - from) : 0;
    // (a line in the absorbing state is decided: nothing it still holds can change its state or a register -- TL_ABSORB)
    const uint32_t absorb = blob[TL_ABSORB];
    // (lazy automata, TL_MISS: a line that steps on a transition nobody has computed is left to the thread-list kernels)
    const uint32_t miss = blob[TL_MISS];
    for (uint32_t p0 = 0; p0 < total && state != 0 && state != absorb && state != miss; p0 += 16) {
        const uint4 v = *q++;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) {
            const uint32_t bi = p0 + j;
            if (bi >= head && bi < total && state != 0 && state != absorb && state != miss) {
                const uint32_t t = trans[state * ncls + cmap[(w[j >> 2] >> ((j & 3) * 8)) & 0xFFu]];
                const uint32_t prog = t >> 16;
                if (prog) {
                    const uint32_t pos = from + (bi - head);  // offset inside the line
                    uint32_t at = opsStart[prog];
                    const uint32_t n = ops[at];
                    for (uint32_t k = 0; k < n; ++k) {
                        const uint32_t op = ops[++at];
                        const uint32_t src = op >> 8;
                        regs[(op & 0xFFu) * kTdfaL2Block + tid] = src == TD_REG_POS ? pos : regs[src * kTdfaL2Block + tid];
                    }
                }
                state = t & 0xFFFFu;
            }
        }
    }
    if (miss && state == miss) {
        status[line] = 4;  // LC_PENDING (nfa_decide_kernel.hpp): the thread-list kernels of the same launch take the line
        if (pendingFlag) atomicMax(pendingFlag, launchSeq);
        return;
    }
    const uint16_t* finalId = reinterpret_cast<const uint16_t*>(base + blob[TL_OFF_FINALID]);
    const uint8_t* finalMap = base + blob[TL_OFF_FINALMAP];
    const uint32_t fid = state ? uint32_t(finalId[state]) : 0xFFFFu;
    const bool matched = fid != 0xFFFFu;
    int32_t* out = caps + size_t(line) * 2 * nGroupsOut;
    for (uint32_t s = 0; s < 2 * nGroupsOut; ++s) {
        int32_t val = -1;
        if (matched && s < nSlots) {
            const uint32_t m = finalMap[fid * nSlots + s];
            if (m == TD_REG_POS) val = int32_t(L);
            else if (m != TD_REG_NONE) val = int32_t(regs[m * kTdfaL2Block + tid]);
        }
        out[s] = val;
    }
    status[line] = matched ? LC_MATCH : LC_NOMATCH;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// tdfa_wave_kernel -- the same automaton, ONE VALUE PER WAVEFRONT: what small and medium batches want (round 4).
//
// A batch of a few thousand values (a Grok entry's candidates, an event group) leaves most of the chip idle whatever the kernel, and
// what it waits for is its LONGEST value.  One value per lane makes that 4096 dependent table reads through L2 (0.7 ms for a 4 KiB
// value, 2.1 ms when the rows come from the Infinity Cache), every lane of the wavefront waiting for the slowest one at every byte.
// Here the state is wave-uniform (table reads are single requests, register programs run once), and the 64 lanes earn their keep on
// the part of a log line that is NOT structure: a byte whose transition is "stay, stamp nothing" starts a run of such bytes in
// practice (GREEDYDATA, a quoted string, a path), and all lanes look for the end of the run at once -- 4 bytes each of the 256-byte
// chunk the wave holds, against the state's QUIET mask (TL_OFF_QUIET: bit c = class c stays in the state without a program).  A
// log line is ~150 bytes of structure and the rest runs; the absorbing state (TL_ABSORB) ends the walk.
// Semantics: exactly tdfa_l2_kernel's (same blob, same results -- tests/test_gpu_parity.py runs both on the same lines).
// LDS: cmap[256] | per wave: registers [nRegs] | (stageBytes != 0) the register programs.

// LT (round 5): the automaton is small (transition table + register programs <= 48 KB: the LDS-size automata of handles that ASK for
// this kernel -- lcPreferWaveTdfa, the Grok matcher's entries): the transition table is staged too, and a byte that is not part of a
// quiet run costs an LDS read instead of a read through L2 (a search wrapper's lazy prefix makes EVERY byte such a byte: 146 -> ~40 ns).
// stageBytes then counts from TL_OFF_TRANS (trans, opsStart, ops are contiguous in the blob).
// (round 6) scalar loads for the walk of tdfaWaveBody<false>: the table word of a wave-uniform (state, class) through the scalar cache
// into an SGPR.  The compiler's own choice for these reads -- the pointers come out of a job table, so their address space is unknown
// to it -- was flat_load_dword + v_readfirstlane with the state and the position in VGPRs and every loop condition an exec mask:
// 45 instructions per byte (profiles/round6_wave_walk_isa.md).  The tables are read-only for the kernel's lifetime.
template <typename T>
__device__ __forceinline__ const T* lcUniformPtr(const T* p) {  // (said to be wave-uniform: both halves through v_readfirstlane)
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(v)), hi = __builtin_amdgcn_readfirstlane(uint32_t(v >> 32));
    return reinterpret_cast<const T*>((uint64_t(hi) << 32) | lo);
}
__device__ __forceinline__ uint32_t lcScalarLoad32(const void* base, uint32_t byteOff) {
    uint32_t v;
    asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(base), "s"(byteOff) : "memory");
    return v;
}
__device__ __forceinline__ uint64_t lcScalarLoad64(const void* base, uint32_t byteOff) {
    uint64_t v;
    asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(base), "s"(byteOff) : "memory");
    return v;
}
__device__ __forceinline__ uint32_t lcScalarLoad16(const void* base, uint32_t index) {  // element `index` of a u16 array
    const uint32_t w = lcScalarLoad32(base, (index << 1) & ~3u);
    return (index & 1u) ? (w >> 16) : (w & 0xFFFFu);
}

template <bool LT>
__device__ __forceinline__ void tdfaWaveBody(const uint8_t* __restrict__ data, const uint32_t* __restrict__ off,
                                             const uint32_t* __restrict__ len, uint32_t sepBytes, uint32_t nLines,
                                             const uint32_t* __restrict__ nLinesPtr, const uint32_t* __restrict__ order,
                                             const uint32_t* __restrict__ resume, const uint32_t* __restrict__ blob,
                                             uint32_t nGroupsOut, int32_t* __restrict__ caps,
                                             uint8_t* __restrict__ status, uint32_t stageBytes,
                                             uint32_t* __restrict__ pendingFlag, uint32_t launchSeq, uint32_t missStatus, uint32_t blockId) {
    extern __shared__ uint32_t wregs[];  // [kTdfaWaveValues][nRegs], then (stageBytes != 0) opsStart, ops
    __shared__ uint8_t cmap[256];
    // (wave-uniform values are SAID to be: the compiler takes anything derived from threadIdx or from an LDS read for divergent, kept the
    // state, the position and every loop condition of the walk in VGPRs and steered it with exec masks -- 15 VALU + 13 SALU per byte)
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    cmap[tid] = reinterpret_cast<const uint8_t*>(blob + TL_HEADER_WORDS)[tid];
    const uint32_t nRegs = blob[TL_NREGS], ncls = blob[TL_NCLASSES], nSlots = blob[TL_NSLOTS];
    uint32_t* regs = wregs + wave * nRegs;
    for (uint32_t r = lane; r < nRegs; r += 64) regs[r] = 0xFFFFFFFFu;  // unset = -1
    if (stageBytes) {
        uint32_t* dst = wregs + kTdfaWaveValues * nRegs;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(blob) + blob[LT ? TL_OFF_TRANS : TL_OFF_OPSSTART]);
        for (uint32_t i = tid; i < stageBytes / 4; i += kTdfaWaveBlock) dst[i] = src[i];
    }
    __syncthreads();
    if (nLinesPtr) nLines = *nLinesPtr < nLines ? *nLinesPtr : nLines;
    const uint32_t slot = blockId * kTdfaWaveValues + wave;
    if (slot >= nLines) return;  // wave-uniform; the workgroup does not synchronise again
    const uint32_t line = __builtin_amdgcn_readfirstlane(order ? order[slot] : slot);
    const uint8_t* base = reinterpret_cast<const uint8_t*>(blob);
    const uint2* quietTab = reinterpret_cast<const uint2*>(base + blob[TL_OFF_QUIET]);
    const uint8_t* staged = reinterpret_cast<const uint8_t*>(wregs + kTdfaWaveValues * nRegs);
    const uint32_t* trans;
    const uint32_t* opsStart;
    const uint16_t* ops;
    if constexpr (LT) {
        trans = reinterpret_cast<const uint32_t*>(staged);
        opsStart = reinterpret_cast<const uint32_t*>(staged + (blob[TL_OFF_OPSSTART] - blob[TL_OFF_TRANS]));
        ops = reinterpret_cast<const uint16_t*>(staged + (blob[TL_OFF_OPS] - blob[TL_OFF_TRANS]));
    } else {
        trans = reinterpret_cast<const uint32_t*>(base + blob[TL_OFF_TRANS]);
        opsStart = stageBytes ? reinterpret_cast<const uint32_t*>(staged) : reinterpret_cast<const uint32_t*>(base + blob[TL_OFF_OPSSTART]);
        ops = stageBytes ? reinterpret_cast<const uint16_t*>(staged + (blob[TL_OFF_OPS] - blob[TL_OFF_OPSSTART]))
                         : reinterpret_cast<const uint16_t*>(base + blob[TL_OFF_OPS]);
    }
    const uint32_t o = __builtin_amdgcn_readfirstlane(off[line]);
    const uint32_t L = __builtin_amdgcn_readfirstlane(len ? len[line] : off[line + 1] - o - sepBytes);
    uint32_t state = blob[TL_START], from = 0;
    if (resume) {
        from = __builtin_amdgcn_readfirstlane(resume[line]);
        from = from < L ? from : L;
        if (from)
            state = __builtin_amdgcn_readfirstlane(
                reinterpret_cast<const uint32_t*>(base + blob[TL_OFF_STARTAFTER])[cmap[data[size_t(o) + from - 1]]]);
    }
    const uint32_t absorb = blob[TL_ABSORB];
    const uint32_t miss = blob[TL_MISS];  // (lazy automata: see tdfa_l2_kernel)
    const uintptr_t addr = reinterpret_cast<uintptr_t>(data) + o;
    const uint32_t head = uint32_t(addr & 3);
    const uint32_t* words = reinterpret_cast<const uint32_t*>(addr - head);
    const uint32_t nWords = L ? (head + L + 3) / 4 : 0;
    const uint32_t end = head + L;
    uint32_t chunk = ((head + from) >> 8);  // index of the 256-byte chunk held in curWord
    // (round 5) ... as CLASSES: every lane maps its four bytes once per chunk (independent LDS reads); the walk's dependent chain per
    // byte is then readlane + the table read, and the run scan tests the four classes it already holds.  The step was bound by the
    // latency of its instructions (profiles/round5_wave_step.txt: 350 cycles a byte for 15 VALU + 13 SALU + 1.8 LDS reads, one wave
    // per SIMD with nothing to hide behind); the class lookup was one of the two LDS reads in the chain.
    auto classesOf = [&](uint32_t w4) {
        return uint32_t(cmap[w4 & 0xFFu]) | (uint32_t(cmap[(w4 >> 8) & 0xFFu]) << 8) | (uint32_t(cmap[(w4 >> 16) & 0xFFu]) << 16) |
               (uint32_t(cmap[w4 >> 24]) << 24);
    };
    uint32_t curWord;
    {
        const uint32_t w = (chunk << 6) + lane;
        curWord = classesOf((w < nWords) ? words[w] : 0);
    }
    uint32_t idx = head + from;  // position in the word-aligned view
    if constexpr (!LT) {
        // ---- the walk, every control value in an SGPR (round 6; see lcScalarLoad32).  Same steps as the loop below, byte for byte.
        const uint32_t uNcls = __builtin_amdgcn_readfirstlane(ncls), uAbsorb = __builtin_amdgcn_readfirstlane(absorb),
                       uMiss = __builtin_amdgcn_readfirstlane(miss), uEnd = __builtin_amdgcn_readfirstlane(end),
                       uHead = __builtin_amdgcn_readfirstlane(head), uWords = __builtin_amdgcn_readfirstlane(nWords);
        uint32_t uState = __builtin_amdgcn_readfirstlane(state), uIdx = __builtin_amdgcn_readfirstlane(idx),
                 uChunk = __builtin_amdgcn_readfirstlane(chunk);
        const bool staged = __builtin_amdgcn_readfirstlane(stageBytes) != 0;
        const uint32_t* gOpsStart = lcUniformPtr(reinterpret_cast<const uint32_t*>(base + blob[TL_OFF_OPSSTART]));
        const uint16_t* gOps = lcUniformPtr(reinterpret_cast<const uint16_t*>(base + blob[TL_OFF_OPS]));
        const uint32_t* uTrans = lcUniformPtr(trans);
        const uint2* uQuiet = lcUniformPtr(quietTab);
        const uint32_t* lOpsStart = wregs + kTdfaWaveValues * nRegs;  // (staged: opsStart, then ops, as in the blob)
        const uint32_t opsDelta = __builtin_amdgcn_readfirstlane(blob[TL_OFF_OPS] - blob[TL_OFF_OPSSTART]);
        auto stops = [&](uint32_t st) { return st == 0u || st == uAbsorb || st == uMiss; };
        // (no branch around the load -- a divergent branch inside the loop makes the compiler steer the WHOLE loop with mask registers:
        // a word index beyond the value reads the value's last word and is replaced by zero)
        auto classesAt = [&](uint32_t w) {
            const uint32_t got = words[w < uWords ? w : uWords - 1u];
            return classesOf(w < uWords ? got : 0u);
        };
#ifdef LC_WAVE_DEBUG  // (-DLC_WAVE_DEBUG: what the walk does on long values, one printf per wavefront; profiles/round6_wave_walk_counts.txt)
        uint32_t dbgMoves = 0, dbgRuns = 0, dbgProgs = 0, dbgOps = 0, dbgScans = 0;
#endif
        if (uIdx < uEnd && !stops(uState)) {
            for (;;) {  // (one byte per turn; leaves by break only)
                if ((uIdx >> 8) != uChunk) {
                    uChunk = uIdx >> 8;
                    const uint32_t w = (uChunk << 6) + lane;
                    curWord = classesAt(w);
                }
                const uint32_t wsel = uint32_t(__builtin_amdgcn_readlane(int(curWord), int((uIdx >> 2) & 63u)));
                const uint32_t cls = (wsel >> ((uIdx & 3u) * 8u)) & 0xFFu;
                const uint32_t t = lcScalarLoad32(uTrans, (uState * uNcls + cls) << 2);
                const uint32_t prog = t >> 16, next = t & 0xFFFFu;
                if (prog) {
#ifdef LC_WAVE_DEBUG
                    ++dbgProgs;
#endif
                    const uint32_t pos = uIdx - uHead;
                    if (staged) {
                        uint32_t at = __builtin_amdgcn_readfirstlane(lOpsStart[prog]);
                        const uint16_t* lOps = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(lOpsStart) + opsDelta);
                        const uint32_t n = __builtin_amdgcn_readfirstlane(lOps[at]);
                        for (uint32_t k = 0; k < n; ++k) {  // (every lane stores the same word: one LDS write)
#ifdef LC_WAVE_DEBUG
                            ++dbgOps;
#endif
                            const uint32_t op = __builtin_amdgcn_readfirstlane(lOps[++at]);
                            const uint32_t src = op >> 8;
                            regs[op & 0xFFu] = src == TD_REG_POS ? pos : regs[src];
                        }
                    } else {
                        uint32_t at = lcScalarLoad32(gOpsStart, prog << 2);
                        const uint32_t n = lcScalarLoad16(gOps, at);
                        for (uint32_t k = 0; k < n; ++k) {
                            const uint32_t op = lcScalarLoad16(gOps, ++at);
                            const uint32_t src = op >> 8;
                            regs[op & 0xFFu] = src == TD_REG_POS ? pos : regs[src];
                        }
                    }
                    uState = next;
                    ++uIdx;
                    if (uIdx >= uEnd || uState == 0u) break;
                    continue;
                }
                if (next == uState) {
                    if (stops(uState)) break;
                    // a quiet byte: find the end of the run -- every lane tests its 4 bytes of the chunk, chunk after chunk
#ifdef LC_WAVE_DEBUG
                    ++dbgRuns;
#endif
                    const uint64_t quiet = lcScalarLoad64(uQuiet, uState << 3);
                    uint32_t stop = uEnd;
                    const uint32_t qlo = uint32_t(quiet), qhi = uint32_t(quiet >> 32);
                    for (;;) {
#ifdef LC_WAVE_DEBUG
                        ++dbgScans;
#endif
                        const uint32_t chunkBase = uChunk << 8;
                        // (32-bit tests: the 64-bit shifts and compares the compiler made of `(quiet >> c) & 1` run at a quarter of the rate)
                        uint32_t notQuiet = 0;  // bit j: byte j of the lane's four is of a class outside the mask
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t c = (curWord >> (8 * j)) & 0xFFu;
                            const uint32_t qw = (c & 32u) ? qhi : qlo;
                            const uint32_t q = (c < 64u) ? ((qw >> (c & 31u)) & 1u) : 0u;
                            notQuiet |= (q ^ 1u) << j;
                        }
                        // bytes behind the position and inside the value: j > idx - base and j < end - base
                        const int32_t base = int32_t(chunkBase + lane * 4u);
                        const int32_t lo = int32_t(uIdx) + 1 - base, hi = int32_t(uEnd) - base;
                        const uint32_t loC = uint32_t(lo < 0 ? 0 : lo > 4 ? 4 : lo), hiC = uint32_t(hi < 0 ? 0 : hi > 4 ? 4 : hi);
                        notQuiet &= ((1u << hiC) - 1u) & ~((1u << loC) - 1u);
                        const uint32_t firstHit = notQuiet ? uint32_t(__ffs(int(notQuiet))) - 1u : 4u;
                        const uint64_t hit = __ballot(notQuiet != 0u);
                        if (hit) {
                            const int l = __ffsll((long long)hit) - 1;
                            stop = chunkBase + uint32_t(l) * 4 + uint32_t(__builtin_amdgcn_readlane(int(firstHit), l));
                            break;
                        }
                        if (chunkBase + 256 >= uEnd) break;  // the run reaches the end of the value
                        ++uChunk;
                        const uint32_t w = (uChunk << 6) + lane;
                        curWord = classesAt(w);
                    }
                    uIdx = stop;
                    if (uIdx >= uEnd) break;
                    continue;
                }
                // (the dead state ends the walk at once; the absorbing state and the MISS sink keep themselves on every class without a
                // program -- regex_handle.cpp packTdfaL2Blob, tdfa.cpp -- and end it at the next byte, in the branch above)
#ifdef LC_WAVE_DEBUG
                ++dbgMoves;
#endif
                uState = next;
                ++uIdx;
                if (uIdx >= uEnd || uState == 0u) break;
            }
        }
#ifdef LC_WAVE_DEBUG
        if (lane == 0 && L >= 2500)
            printf("wavewalk L %u walked %u moves %u runs %u scans %u progs %u ops %u staged %u ncls %u nstates %u\n", L, uIdx - uHead, dbgMoves, dbgRuns, dbgScans,
                   dbgProgs, dbgOps, staged ? 1u : 0u, uNcls, blob[TL_NSTATES]);
#endif
        state = uState;
        idx = uIdx;
    } else
    while (idx < end && state != 0 && state != absorb && state != miss) {
        if ((idx >> 8) != chunk) {
            chunk = idx >> 8;
            const uint32_t w = (chunk << 6) + lane;
            curWord = classesOf((w < nWords) ? words[w] : 0);
        }
        const uint32_t wsel = __builtin_amdgcn_readlane(curWord, (idx >> 2) & 63u);
        const uint32_t cls = (wsel >> ((idx & 3u) * 8)) & 0xFFu;
        const uint32_t t = __builtin_amdgcn_readfirstlane(trans[state * ncls + cls]);
        const uint32_t prog = t >> 16, next = t & 0xFFFFu;
        if (prog) {
            const uint32_t pos = idx - head;
            uint32_t at = __builtin_amdgcn_readfirstlane(opsStart[prog]);
            const uint32_t n = __builtin_amdgcn_readfirstlane(ops[at]);
            for (uint32_t k = 0; k < n; ++k) {  // (every lane stores the same word: one LDS write)
                const uint32_t op = __builtin_amdgcn_readfirstlane(ops[++at]);
                const uint32_t src = op >> 8;
                regs[op & 0xFFu] = src == TD_REG_POS ? pos : regs[src];
            }
        } else if (next == state) {
            // a quiet byte: find the end of the run -- every lane tests its 4 bytes of the chunk, chunk after chunk
            const uint2 q2 = quietTab[state];
            const uint64_t quiet = (uint64_t(q2.y) << 32) | q2.x;
            uint32_t stop = end;
            for (;;) {
                const uint32_t chunkBase = chunk << 8;
                uint32_t firstHit = 4;
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    const uint32_t bi = chunkBase + lane * 4 + uint32_t(j);
                    const uint32_t c = (curWord >> (8 * j)) & 0xFFu;
                    const bool isQuiet = c < 64 && ((quiet >> c) & 1ull);
                    if (bi > idx && bi < end && !isQuiet) firstHit = uint32_t(j);
                }
                const uint64_t hit = __ballot(firstHit < 4);
                if (hit) {
                    const int l = __ffsll((long long)hit) - 1;
                    stop = chunkBase + uint32_t(l) * 4 + uint32_t(__builtin_amdgcn_readlane(int(firstHit), l));
                    break;
                }
                if (chunkBase + 256 >= end) break;  // the run reaches the end of the value
                ++chunk;
                const uint32_t w = (chunk << 6) + lane;
                curWord = classesOf((w < nWords) ? words[w] : 0);
            }
            idx = stop;
            continue;
        }
        state = next;
        ++idx;
    }
    if (miss && state == miss) {
        if (lane == 0) {
            status[line] = uint8_t(missStatus);  // LC_PENDING (the handle's own chain), or LC_OVERFLOW (the Grok plan's second chance)
            if (pendingFlag) atomicMax(pendingFlag, launchSeq);
        }
        return;
    }
    const uint16_t* finalId = reinterpret_cast<const uint16_t*>(base + blob[TL_OFF_FINALID]);
    const uint8_t* finalMap = base + blob[TL_OFF_FINALMAP];
    const uint32_t fid = state ? uint32_t(finalId[state]) : 0xFFFFu;
    const bool matched = fid != 0xFFFFu;
    int32_t* out = caps + size_t(line) * 2 * nGroupsOut;
    for (uint32_t s = lane; s < 2 * nGroupsOut; s += 64) {
        int32_t val = -1;
        if (matched && s < nSlots) {
            const uint32_t m = finalMap[fid * nSlots + s];
            if (m == TD_REG_POS) val = int32_t(L);
            else if (m != TD_REG_NONE) val = int32_t(regs[m]);
        }
        out[s] = val;
    }
    if (lane == 0) status[line] = matched ? LC_MATCH : LC_NOMATCH;
}

template <bool LT>
__global__ __launch_bounds__(kTdfaWaveBlock) void tdfa_wave_kernel(const uint8_t* __restrict__ data, const uint32_t* __restrict__ off,
                                                                  const uint32_t* __restrict__ len, uint32_t sepBytes, uint32_t nLines,
                                                                  const uint32_t* __restrict__ nLinesPtr, const uint32_t* __restrict__ order,
                                                                  const uint32_t* __restrict__ resume, const uint32_t* __restrict__ blob,
                                                                  uint32_t nGroupsOut, int32_t* __restrict__ caps,
                                                                  uint8_t* __restrict__ status, uint32_t stageBytes,
                                                                  uint32_t* __restrict__ pendingFlag, uint32_t launchSeq) {
    tdfaWaveBody<LT>(data, off, len, sepBytes, nLines, nLinesPtr, order, resume, blob, nGroupsOut, caps, status, stageBytes, pendingFlag,
                     launchSeq, 4u /* LC_PENDING */, blockIdx.x);
}

// ---- several automata, each over its OWN values, in ONE launch (round 6).  Round 0 of a Grok batch is ~46 of these walks, one per
// Match entry over that entry's few hundred candidates: each a launch of 30-100 us that the host needs 15 us to queue, plus a post
// launch per entry -- the round lasted as long as queueing its launches did (profiles/round6_grok_timeline_lazy.txt: the 46th kernel
// started 1.4 ms after the first).  Here workgroup b belongs to the job whose [firstBlock, next firstBlock) holds b (a binary search
// over at most 64 words in device memory), takes that job's tables, values and outputs, and walks as tdfa_wave_kernel does.
// Jobs of lazy automata (TL_MISS) name the flag word and the status their misses raise.
// jobTable: u32 firstBlock[kTdfaWaveMaxJobs] (ascending; unused = 0xFFFFFFFF), then TdfaWaveJob[nJobs] (8-byte aligned)
__global__ __launch_bounds__(kTdfaWaveBlock) void tdfa_wave_multi_kernel(const uint8_t* __restrict__ data, const uint32_t* __restrict__ jobTable,
                                                                        uint32_t nJobs) {
    const uint32_t b = blockIdx.x;
    uint32_t lo = 0, hi = nJobs;  // the last job whose firstBlock <= b
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (jobTable[mid] <= b) lo = mid;
        else hi = mid;
    }
    const TdfaWaveJob j = reinterpret_cast<const TdfaWaveJob*>(jobTable + kTdfaWaveMaxJobs)[lo];  // (wave-uniform: scalar loads)
    tdfaWaveBody<false>(data, j.off, j.len, 0u, j.n, nullptr, nullptr, j.resume, j.blob, j.nGroupsOut, j.caps, j.status, j.stageBytes, j.missFlag,
                        j.seq, j.missStatus, b - j.firstBlock);
}
