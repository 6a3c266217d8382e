// tdfa_stream_kernel.hpp -- the tagged-DFA engine with its LDS instructions INTERLEAVED per byte (included by
// gpu_runtime.hip only; tdfa_kernel.hpp holds the phase-separated original, which stays for the byte-pair and byte-row
// table formats).
//
// Why a second kernel.  tdfa_match_kernel steps a 16-byte chunk in three bursts: 16 class lookups, the 16 dependent chain
// links, 16 capture stamps.  The CU's LDS serves the instructions of all its waves in arrival order, so a chain link -- the
// only thing a wave ever waits for -- queues behind the 32..64 LDS cycles of whatever burst another wave has just issued:
// measured, a link costs ~310 cycles in that kernel against ~115 for the same mix issued one-one-one
// (tools/lds_chain_bench.hip mode 2), with the LDS only half busy (profiles/round1: SQ_LDS_IDX_ACTIVE 55 %).  Here every
// step issues exactly
//      the chain link of byte j of THIS chunk          t = lds32[(t & 0xFFFF) + col[j]]        (waited for by step j+1)
//      the class lookup of byte j of the NEXT chunk    ncol[j] = cmap8[next byte j]            (waited for a chunk later)
//      the capture stamp of byte j of the PREVIOUS one regs[ptt[j] >> 16][lane] = pos          (never waited for)
// so the queue in front of a link holds at most two cheap instructions per resident wave, and nothing of the wave's own.
//
// The stamps trail by a whole chunk (not by a byte) because of the general register programs (tdfa_kernel.hpp): a chunk in
// which some lane met one is replayed in order, which is only correct if none of its simple stamps has been issued yet.
// After a replay the chunk's trailing stamps are pointed at the dummy register.
//
// Staging is the original's (cooperative 64-byte stages through a per-wave LDS tile), one stage further ahead: during stage
// s the tile already holds stage s+1, because the class lookups of a stage's first chunk are issued during the last chunk of
// the stage before.  W[k] holds the words of chunk k: of stage s until chunk k has been stepped, of stage s+1 from then on.
#pragma once

#include "tdfa_kernel.hpp"

#ifndef LC_TDFA_STREAM_CHUNK
#define LC_TDFA_STREAM_CHUNK 8  // bytes per chunk (8 or 16): col / ncol / ptt / tt hold one value per byte of a chunk
#endif
#ifndef LC_TDFA_GCLASS
#define LC_TDFA_GCLASS 0  // 1: the one-stamp pair kernel reads the second byte's class through the vector L1 instead of LDS (experiment)
#endif
#ifndef LC_TDFA_STAMP_ISA
#define LC_TDFA_STAMP_ISA 1  // hand-picked instructions for the one-stamp pair kernel's stamp (0: the compiler's)
#endif
#ifndef LC_TDFA_ROW_ALIGN
#define LC_TDFA_ROW_ALIGN 1
#endif
#ifndef LC_TDFA_STREAM_WAVES
#define LC_TDFA_STREAM_WAVES 4  // waves per SIMD the register allocator must leave room for (128 VGPRs)
#endif

// LDS byte address of the low byte of hdr[TD_ID_COL]: a class "lookup" of a byte outside the line reads the identity
// column's offset from there (identity column < 256: at most 63 classes), so the checked path has no select after the read
constexpr uint32_t kTdfaIdColByteAddr = TD_ID_COL * 4;

// steps the NB (8 or 16) bytes whose classes are in col[]; nwords = the NB bytes that follow them
template <int NB, bool CHECKED, typename TdfaReg, int LAB>
__device__ __forceinline__ uint32_t tdfaStreamChunk(uint32_t t, const uint32_t (&col)[NB], uint32_t (&ncol)[NB],
                                                    const uint32_t (&nwords)[NB / 4], uint32_t nbase, uint32_t L, uint32_t idCol,
                                                    const uint32_t (&ptt)[NB], uint32_t (&tt)[NB], uint32_t pbase,
                                                    uint32_t regAddr0, uint32_t& seenOut) {
    typedef LdsRegPtrT<TdfaReg> LdsRegPtr;
    uint32_t seen = 0;
    // (the byte extraction and the running position are volatile asm: left to itself the compiler computes all NB of each
    // ahead of the chunk -- 2*NB registers for values that cost one VALU instruction in the shadow of an LDS round trip)
    uint32_t pos = pbase;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        t = *reinterpret_cast<LdsWordPtr>(addLowHalf(col[j], t));
        tt[j] = t;
        uint32_t b;
        asm volatile("v_bfe_u32 %0, %1, %2, 8" : "=v"(b) : "v"(nwords[j >> 2]), "n"((j & 3) * 8));
        if constexpr ((LAB & kLabPreClass) != 0) {
            ncol[j] = CHECKED ? ((nbase + j < L) ? b : idCol) : b;
        } else if constexpr (CHECKED) {
            ncol[j] = *reinterpret_cast<LdsBytePtr>((nbase + j < L) ? TD_CMAP_OFFSET + b : kTdfaIdColByteAddr);
        } else {
            ncol[j] = *reinterpret_cast<LdsBytePtr>(TD_CMAP_OFFSET + b);
        }
        if constexpr ((LAB & kLabNoStamp) == 0) {
            *reinterpret_cast<LdsRegPtr>(addHighHalf(regAddr0, ptt[j])) = TdfaReg(pos);
            asm volatile("v_add_u32 %0, 1, %0" : "+v"(pos));
        }
        if constexpr ((LAB & kLabNoGeneral) == 0)
            if (j > 0) seen |= tt[j - 1];  // (the link before this one: already waited for)
#ifndef LC_TDFA_STREAM_NO_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);  // keep the one-one-one order: the scheduler would cluster the reads and sink the stores
#endif
    }
    seenOut = seen | t;
    return t;
}

// Byte-PAIR tables (device_tables.h TP_*): one chain link per TWO bytes.  The chain is what a wave waits for (16 waves per CU
// x one link in flight each, ~120 cycles per link), so halving the links per byte is worth more here than in the
// phase-separated kernel, where it bought nothing.  Per pair: the link, the two class lookups of the
// NEXT chunk's pair (first byte: cmapA, u16; second byte: cmap8), the two stamps of the PREVIOUS chunk's pair.
template <int BLOCK, int NB, bool CHECKED, typename TdfaReg, int LAB = 0>
__device__ __forceinline__ uint32_t tdfaStreamPairChunk(uint32_t t, const uint32_t (&colp)[NB / 2], uint32_t (&na)[NB / 2],
                                                        uint32_t (&nc)[NB / 2], const uint32_t (&nwords)[NB / 4], uint32_t nbase,
                                                        uint32_t L, uint32_t cmapA, uint32_t idAAddr,
                                                        const uint32_t (&ptt)[NB / 2], uint32_t (&tt)[NB / 2], uint32_t pbase,
                                                        uint32_t regAddr0, uint32_t& seenOut) {
    typedef LdsRegPtrT<TdfaReg> LdsRegPtr;
    constexpr uint32_t kRegShift = (BLOCK == 1024 ? 12 : BLOCK == 512 ? 11 : BLOCK == 256 ? 10 : BLOCK == 128 ? 9 : 8) -
                                   (sizeof(TdfaReg) == 2 ? 1 : 0);  // log2(BLOCK * sizeof(TdfaReg))
    uint32_t seen = 0;
    uint32_t pos = pbase;
#pragma unroll
    for (int p = 0; p < NB / 2; ++p) {
        t = *reinterpret_cast<LdsWordPtr>(addLowHalf(colp[p], t));
        tt[p] = t;
        uint32_t b0, b1;
        asm volatile("v_bfe_u32 %0, %1, %2, 8" : "=v"(b0) : "v"(nwords[p >> 1]), "n"((p & 1) * 16));
        asm volatile("v_bfe_u32 %0, %1, %2, 8" : "=v"(b1) : "v"(nwords[p >> 1]), "n"((p & 1) * 16 + 8));
        if constexpr (CHECKED) {
            na[p] = *reinterpret_cast<LdsHalfPtr>((nbase + 2 * p < L) ? cmapA + b0 * 2 : idAAddr);
            nc[p] = *reinterpret_cast<LdsBytePtr>((nbase + 2 * p + 1 < L) ? TD_CMAP_OFFSET + b1 : kTdfaIdColByteAddr);
        } else {
            na[p] = *reinterpret_cast<LdsHalfPtr>(cmapA + b0 * 2);
            nc[p] = *reinterpret_cast<LdsBytePtr>(TD_CMAP_OFFSET + b1);
        }
        if constexpr ((LAB & kLabOneStamp) != 0) {
            // (timing only, profiles/round3_tdfa_experiments.txt: what ONE stamp per pair would cost.  A register row that exists:
            // the real thing would be one SDWA add on a pre-scaled field, this is two instructions)
            *reinterpret_cast<LdsRegPtr>(regAddr0 + (((ptt[p] >> 16) & 0xFu) << kRegShift)) = TdfaReg(pos);
            asm volatile("v_add_u32 %0, 2, %0" : "+v"(pos));
        } else {
            const uint32_t r0 = (ptt[p] >> 16) & 0xFFu, r1 = ptt[p] >> 24;
            *reinterpret_cast<LdsRegPtr>(regAddr0 + (r0 << kRegShift)) = TdfaReg(pos);
            asm volatile("v_add_u32 %0, 1, %0" : "+v"(pos));
            *reinterpret_cast<LdsRegPtr>(regAddr0 + (r1 << kRegShift)) = TdfaReg(pos);
            asm volatile("v_add_u32 %0, 1, %0" : "+v"(pos));
        }
        if constexpr ((LAB & kLabNoGeneral) == 0)
            if (p > 0) seen |= tt[p - 1];
#ifndef LC_TDFA_STREAM_NO_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    seenOut = seen | t;
    return t;
}

// ONE-STAMP pair tables (device_tables.h TP1_*, regex_handle.cpp: LC_TDFA_PAIR=2).  A pair entry names ONE register (rA) and
// whether it takes the position of the pair's first or second byte; the capture stamp -- the dearest LDS instruction of the walk
// (4 cycles: address + data) -- is issued once per TWO bytes.  Registers that are always stamped one byte behind another one (end
// of field k on the separator, start of field k+1 behind it: every line, several times) have no stamps of their own (derive words,
// applied with the results).  What is left are DOUBLE entries (a one-byte field: begin and end stamped in consecutive bytes that
// happen to share a pair): rA is the first byte's register, and the second one's (rB) is settled behind the chunk's stamps as
// max(register, pos + 1) -- only in chunks in which some lane of the wavefront met a DOUBLE.  Positions only grow along a line
// and registers start at 0, so "latest" is "largest" and the order of the two kinds of store among themselves does not matter:
// tests/helpers/table_interp.py TdfaPair1Interp is this function, store for store.
template <int BLOCK, int NB, bool CHECKED, typename TdfaReg, bool A8 = false>
__device__ __forceinline__ uint32_t tdfaStreamPair1Chunk(uint32_t t, const uint32_t (&colp)[NB / 2], uint32_t (&na)[NB / 2],
                                                         uint32_t (&nc)[NB / 2], const uint32_t (&nwords)[NB / 4], uint32_t nbase,
                                                         uint32_t L, uint32_t cmapA, uint32_t idAAddr,
                                                         const uint32_t (&ptt)[NB / 2], uint32_t (&tt)[NB / 2], uint32_t pbase,
                                                         uint32_t regAddr0, const uint8_t* __restrict__ gcmap8 = nullptr) {
    typedef LdsRegPtrT<TdfaReg> LdsRegPtr;
    constexpr uint32_t kRegShift = (BLOCK == 1024 ? 12 : BLOCK == 512 ? 11 : BLOCK == 256 ? 10 : BLOCK == 128 ? 9 : 8) -
                                   (sizeof(TdfaReg) == 2 ? 1 : 0);  // log2(BLOCK * sizeof(TdfaReg))
#pragma unroll
    for (int p = 0; p < NB / 2; ++p) {
        t = *reinterpret_cast<LdsWordPtr>(addLowHalf(colp[p], t));
        tt[p] = t;
        uint32_t b0, b1;
        asm volatile("v_bfe_u32 %0, %1, %2, 8" : "=v"(b0) : "v"(nwords[p >> 1]), "n"((p & 1) * 16));
        asm volatile("v_bfe_u32 %0, %1, %2, 8" : "=v"(b1) : "v"(nwords[p >> 1]), "n"((p & 1) * 16 + 8));
        if constexpr (CHECKED) {
            if constexpr (A8) na[p] = *reinterpret_cast<LdsBytePtr>((nbase + 2 * p < L) ? cmapA + b0 : idAAddr);  // (cmapA / idAAddr: the u8 copy, its 257th byte)
            else na[p] = *reinterpret_cast<LdsHalfPtr>((nbase + 2 * p < L) ? cmapA + b0 * 2 : idAAddr);
            nc[p] = *reinterpret_cast<LdsBytePtr>((nbase + 2 * p + 1 < L) ? TD_CMAP_OFFSET + b1 : kTdfaIdColByteAddr);
        } else {
            if constexpr (A8) na[p] = *reinterpret_cast<LdsBytePtr>(cmapA + b0);
            else na[p] = *reinterpret_cast<LdsHalfPtr>(cmapA + b0 * 2);
#if LC_TDFA_GCLASS
            // Round 6 experiment: the SECOND byte's class from the table's copy in global memory (256 bytes: two cache lines that never
            // leave the vector L1) -- the texture path is idle in this kernel, the LDS queue is what the chain link waits in
            // (profiles/round6_tdfa_why_not.md): one LDS instruction of four per byte pair moves off it.
            nc[p] = gcmap8[b1];
#else
            nc[p] = *reinterpret_cast<LdsBytePtr>(TD_CMAP_OFFSET + b1);
#endif
        }
        // the previous chunk's pair p: ONE stamp
#if LC_TDFA_STAMP_ISA
        // Round 6: the stamp's address and value in FOUR instructions instead of the six the compiler picks (v_lshrrev 6, v_and
        // 0x1fc00, v_add | v_lshrrev 23, v_and_or, v_or: profiles/round5_tdfa_isa_budget.md, 3.25 of the kernel's 7.47 VALU per line
        // byte): register index and +1 flag by v_bfe_u32, row address by v_lshl_add_u32, and the pair's position -- wave-uniform
        // when rows start at the line's first byte -- added on the scalar unit.
        if constexpr (LC_TDFA_ROW_ALIGN == 1) {
            uint32_t ra, delta, addr, val;
            asm("v_bfe_u32 %0, %1, 16, 7" : "=v"(ra) : "v"(ptt[p]));
            asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(addr) : "v"(ra), "n"(kRegShift), "v"(regAddr0));
            asm("v_bfe_u32 %0, %1, 23, 1" : "=v"(delta) : "v"(ptt[p]));
            const uint32_t posU = __builtin_amdgcn_readfirstlane(pbase) + uint32_t(2 * p);
            asm("v_add_u32 %0, %1, %2" : "=v"(val) : "s"(posU), "v"(delta));
            *reinterpret_cast<LdsRegPtr>(addr) = TdfaReg(val);
        } else
#endif
        {
            const uint32_t ra = (ptt[p] >> 16) & 0x7Fu;
            *reinterpret_cast<LdsRegPtr>(regAddr0 + (ra << kRegShift)) = TdfaReg(pbase + uint32_t(2 * p) + ((ptt[p] >> 23) & 1u));
        }
#ifndef LC_TDFA_STREAM_NO_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    return t;
}
// the second registers of the DOUBLE entries among ptt[] (the pairs at line offset pbase, pbase + 2, ...): behind their chunk's stamps
template <int BLOCK, int NP, typename TdfaReg>
__device__ __forceinline__ void tdfaSettleDoubles(const uint32_t (&ptt)[NP], uint32_t pbase, uint32_t regAddr0) {
    typedef LdsRegPtrT<TdfaReg> LdsRegPtr;
    constexpr uint32_t kRegShift = (BLOCK == 1024 ? 12 : BLOCK == 512 ? 11 : BLOCK == 256 ? 10 : BLOCK == 128 ? 9 : 8) -
                                   (sizeof(TdfaReg) == 2 ? 1 : 0);
    uint32_t any = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) any |= ptt[p];
    if (!__any(int32_t(any) < 0)) return;  // (the usual case: no lane of the wavefront met a DOUBLE in this chunk)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (int32_t(ptt[p]) < 0) {
            const LdsRegPtr reg = reinterpret_cast<LdsRegPtr>(regAddr0 + (((ptt[p] >> 24) & 0x7Fu) << kRegShift));
            const TdfaReg val = TdfaReg(pbase + uint32_t(2 * p) + 1u);
            const TdfaReg cur = *reg;
            *reg = cur > val ? cur : val;
        }
    }
}

// ---- LDS-DMA staging (kLabDmaStage; COMPACT tiles only: rows of exactly 64 bytes).  global_load_lds_dwordx4 writes the 16 bytes
// of lane l at (wave-uniform LDS base in M0) + 16 * l, so one instruction fills 16 rows of the tile: lane l carries position
// l % 4 of row l / 4.  The tile is XOR-swizzled (segment g of row r sits at position g ^ ((r >> 1) & 3)), and the destination
// cannot be permuted, so the SOURCE is: the lane at position p fetches segment p ^ ((r >> 1) & 3).  No staging VGPRs, no
// ds_write_b128 (13 LDS cycles each, MI355X_MICROARCH.md) -- the tile itself is the buffer, so a stage's DMA is issued when the
// row of the stage before has been read into W completely, and waited for (vmcnt, the issuing wave's own counter: nobody else
// reads this wave's tile) before W is refilled.  Lanes outside their row's span issue nothing: the tile keeps stale bytes there,
// which no line reads as payload (bytes outside the line take the identity column).
__device__ __forceinline__ void tdfaDmaLoad16(uint32_t ldsDst, uint64_t base, uint32_t off) {
    uint32_t keep;  // (M0 belongs to the compiler: saved and restored inside the statement that uses it)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(off), "s"(ldsDst), "s"(base)
                 : "memory");
}
__device__ __forceinline__ void tdfaDmaWait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void tdfaLdsDrain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// the kernel body; blockId = index of this workgroup among the workgroups of ITS batch (tdfa_stream_multi_kernel packs the
// workgroups of several batches, each with its own tables, into one launch)
template <int BLOCK, bool COMPACT, bool PAIR, int LAB>
__device__ __forceinline__ void tdfaStreamBody(
    const uint8_t* __restrict__ data, const uint32_t* __restrict__ off, const uint32_t* __restrict__ len, uint32_t sepBytes,
    uint32_t minLen, uint32_t nLines, const uint32_t* __restrict__ nLinesPtr, const uint32_t* __restrict__ order,
    const uint32_t* __restrict__ resume, const uint32_t* __restrict__ blob, uint32_t blobBytes, uint32_t regBytes,
    uint32_t nGroupsOut, int32_t* __restrict__ caps, uint8_t* __restrict__ status, uint32_t* __restrict__ longFlag,
    uint32_t launchSeq, uint32_t blockId) {
    static_assert(kTdfaStageBytes == 64 || kTdfaStageBytes == 128, "4 or 8 16-byte segments per stage");
    constexpr int kLoads = kTdfaLoads;  // 16-byte segments per staged row == lanes that share one line == loads per stage
    static_assert(!PAIR || (LAB & kLabPreClass) == 0, "no pre-classified pairs");
    typedef typename std::conditional<COMPACT, uint16_t, uint32_t>::type TdfaReg;
    typedef LdsRegPtrT<TdfaReg> LdsRegPtr;
    constexpr uint32_t kRowStride = COMPACT ? kTdfaStageBytes : kTdfaRowStride;  // (tdfa_match_kernel's two tile layouts)
    constexpr uint32_t kStagePerWave = 64 * kRowStride;
    constexpr bool DMA = COMPACT && (LAB & kLabDmaStage) != 0 && kTdfaStageBytes == 64;
    constexpr bool PAIR1 = PAIR && (LAB & kLabPairOne) != 0;  // the pair table is a ONE-STAMP table (the launcher checks TP_FORMAT)
    static_assert(!PAIR1 || (LAB & kLabNoGeneral) != 0, "one-stamp pair tables have no general register programs");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    if (nLinesPtr) {
        const uint32_t dyn = *nLinesPtr;
        nLines = dyn < nLines ? dyn : nLines;
    }
    if (minLen && longFlag && __atomic_load_n(longFlag, __ATOMIC_RELAXED) < launchSeq) return;
    if (minLen) {  // mop-up launch behind a COMPACT one (see tdfa_match_kernel)
        const uint32_t s0 = blockId * BLOCK + tid;
        bool mine = false;
        if (s0 < nLines) {
            const uint32_t ln = order ? order[s0] : s0;
            uint32_t l0 = len ? len[ln] : off[ln + 1] - off[ln] - sepBytes;
            if (resume) {
                const uint32_t f0 = resume[ln];
                l0 -= f0 < l0 ? f0 : l0;
            }
            mine = l0 >= minLen;
        }
        volatile uint32_t* flag = reinterpret_cast<volatile uint32_t*>(smem);
        if (tid == 0) *flag = 0;
        __syncthreads();
        if (mine) *flag = 1;
        __syncthreads();
        const bool any = *flag != 0;
        __syncthreads();
        if (!any) return;
    }
    {
        const uint4* src = reinterpret_cast<const uint4*>(blob);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = tid; i < blobBytes / 16; i += BLOCK) dst[i] = src[i];
        tdfaClearRegisters<BLOCK>(smem, blob, blobBytes, regBytes);
        if constexpr (PAIR1 && (LAB & kLabCmapA8) != 0) {
            // the u8 copy of the first-byte class map, behind the tiles: class INDEX = cmapA[b] / (the first byte's stride); byte 256 = the
            // identity class (a byte outside the line)
            const uint32_t po = blob[TD_OFF_PAIR];
            const uint32_t* gph = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(blob) + po);
            const uint16_t* gA = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(blob) + gph[TP_OFF_CMAPA]);
            const uint32_t strideA = blob[TD_ID_COL] + 4;  // (classes + 1) * 4
            uint8_t* a8 = smem + blobBytes + regBytes + (BLOCK / 64) * 64 * (COMPACT ? kTdfaStageBytes : kTdfaRowStride);
            for (uint32_t i = tid; i < 257; i += BLOCK) a8[i] = uint8_t((i < 256 ? uint32_t(gA[i]) : gph[TP_ID_A]) / strideA);
        }
        if constexpr (PAIR1) {  // ("latest stamp" = "largest offset" for the settled doubles: every register starts at 0)
            if (((blob[TD_NREGS] >> 16) & 0x1FFFu) == 0) {
                uint32_t* regs = reinterpret_cast<uint32_t*>(smem + blobBytes);
                for (uint32_t i = threadIdx.x; i < regBytes / 4; i += BLOCK) regs[i] = 0;
            }
        }
    }
    __syncthreads();
    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(smem);
    const uint32_t rowBytes = hdr[TD_ROW_BYTES];
    const uint32_t idCol = hdr[TD_ID_COL];
    const uint32_t regsBase = blobBytes;
    // a transition that stamps the dummy register (the last one) and nothing else
    const uint32_t dummyT = PAIR1 ? ((hdr[TD_NREGS] & 0xFFFFu) - 1) << 16 : PAIR ? ((hdr[TD_NREGS] & 0xFFFFu) - 1) * 0x01010000u : (((hdr[TD_NREGS] & 0xFFFFu) - 1) * BLOCK * uint32_t(sizeof(TdfaReg))) << 16;
    uint32_t t = hdr[TD_START_ROW];
    TdfaPairInfo pi{};
    uint32_t idAAddr = 0;  // LDS address of a u16 that holds pi.idA (the first-byte offset of the identity class)
    // single-byte row address -> the row the kernel walks (PAIR: the same state's row in the pair table)
    auto walkRow = [&](uint32_t singleRow) {
        return PAIR ? pi.base + ((singleRow & 0xFFFFu) - TD_TRANS_OFFSET) / rowBytes * pi.rowBytes : singleRow;
    };
    if constexpr (PAIR) {
        const uint32_t* ph = reinterpret_cast<const uint32_t*>(smem + hdr[TD_OFF_PAIR]);
        pi = TdfaPairInfo{ph[TP_BASE], ph[TP_ROW_BYTES], ph[TP_OFF_CMAPA], ph[TP_ID_A]};
        idAAddr = hdr[TD_OFF_PAIR] + TP_ID_A * 4;
        t = walkRow(t);
    }
    const uint32_t deadRow = PAIR ? pi.base : TD_TRANS_OFFSET;

    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t stageBase = blobBytes + regBytes + wave * kStagePerWave;
    // (kLabCmapA8) LDS address of the u8 first-byte class map and the stride a class index scales by
    const uint32_t cmapA8 = blobBytes + regBytes + (BLOCK / 64) * kStagePerWave;
    const uint32_t strideA8 = idCol + 4;

    const uint32_t tStart = t;
    // ---- one block of BLOCK lines (everything from here on is the wavefront's own: its tile, its lanes' register columns)
    auto walkBlock = [&](const uint32_t blockNow) {
    t = tStart;
    const uint32_t slot = blockNow * BLOCK + tid;
    bool live = slot < nLines;
    const uint32_t line = (live && order) ? order[slot] : slot;
    uint32_t o = 0, L = 0;
    uint32_t from = 0;
    if (live) {
        o = off[line];
        L = len ? len[line] : off[line + 1] - o - sepBytes;
        if (resume) {
            from = resume[line];
            from = from < L ? from : L;
            if (from) {
                const uint32_t* startAfter = reinterpret_cast<const uint32_t*>(smem + hdr[TD_OFF_STARTAFTER]);
                t = walkRow(startAfter[smem[TD_CMAP_OFFSET + data[size_t(o) + from - 1]] >> 2]);
                o += from;
                L -= from;
            }
        }
        if ((COMPACT && L > kTdfaWideMaxLine) || L < minLen) {
            if (COMPACT && L > kTdfaWideMaxLine && longFlag) atomicMax(longFlag, launchSeq);
            live = false;
            L = 0;
        }
    }
    const uintptr_t addr = reinterpret_cast<uintptr_t>(data) + o;
    // Round 5: a line's tile rows start AT its first byte (LC_TDFA_ROW_ALIGN = 1: head = 0).  Through round 4 they started at the
    // 16-byte boundary below it, so that every 16-byte load was aligned -- and 15 of 16 lines began with junk bytes in their first two
    // chunks (the copy of the chunk loop that tests every byte) and needed one stage more than their length: a 512-byte line took 9
    // stages, the ninth for its last `head` bytes.  gfx950 takes the misaligned 16-byte loads (global_load_dwordx4 and the LDS-DMA form
    // alike: ROCm runs compute queues in unaligned-access mode) at no measurable cost: 512-byte lines 0.1816 -> 0.1720 ms per Mi lines,
    // roofline.frac 0.433 -> 0.458; the parity suites (ragged, unaligned, empty, 64 KiB+ lines) pass either way.  -DLC_TDFA_ROW_ALIGN=16
    // restores the aligned rows.
#ifndef LC_TDFA_ROW_ALIGN
#define LC_TDFA_ROW_ALIGN 1
#endif
    static_assert(LC_TDFA_ROW_ALIGN == 1 || LC_TDFA_ROW_ALIGN == 4 || LC_TDFA_ROW_ALIGN == 16, "row alignment");
    const uint32_t head = uint32_t(addr & (LC_TDFA_ROW_ALIGN - 1));
    const uintptr_t rowStart = addr - head;
    const uint32_t span = L ? head + L : 0;
    const uint32_t myStages = (span + kTdfaStageBytes - 1) / kTdfaStageBytes;
    uint32_t maxStages = myStages;
    // the wave's largest head and smallest head + length (a lane without a line makes the wave "never full": its bytes are not a line's)
    uint32_t waveMaxHead = head, waveMinSpan = live ? span : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t other = __shfl_xor(maxStages, d, 64);
        maxStages = other > maxStages ? other : maxStages;
        const uint32_t oh = __shfl_xor(waveMaxHead, d, 64), os = __shfl_xor(waveMinSpan, d, 64);
        waveMaxHead = oh > waveMaxHead ? oh : waveMaxHead;
        waveMinSpan = os < waveMinSpan ? os : waveMinSpan;
    }
    waveMaxHead = __builtin_amdgcn_readfirstlane(waveMaxHead);
    waveMinSpan = __builtin_amdgcn_readfirstlane(waveMinSpan);

    const uint32_t seg = (lane % kLoads) * 16;
    // loads are addressed as (16-byte aligned buffer base, wave-uniform) + a 32-bit offset per lane: half the address registers
    const uintptr_t dataAligned = reinterpret_cast<uintptr_t>(data) & ~uintptr_t(15);
    const uint32_t rowOff = uint32_t(rowStart - dataAligned);  // (line offsets are 32-bit)
    uint32_t srcOff[kLoads], srcSpan[kLoads], dstAddr[kLoads];
    // DMA: position (lane % kLoads) of the row carries source segment position ^ swizzle(row); (row >> 1) & 3 == (lane >> 3) & 3
    // for every one of the kLoads instructions (their rows are 16 apart)
    const uint32_t dmaSeg = DMA ? (((lane % kLoads) ^ ((lane >> 3) & uint32_t(kLoads - 1))) << 4) : seg;
#pragma unroll
    for (int i = 0; i < kLoads; ++i) {
        const int r = (64 / kLoads) * i + int(lane / kLoads);
        srcOff[i] = __shfl(rowOff, r, 64) + dmaSeg;
        srcSpan[i] = __shfl(span, r, 64);
        dstAddr[i] = stageBase + uint32_t(r) * kRowStride + (COMPACT ? seg ^ (((uint32_t(r) >> 1) & uint32_t(kLoads - 1)) << 4) : seg);
    }
    const uint32_t myRow = stageBase + lane * kRowStride;
    const uint32_t mySwizzle = COMPACT ? ((lane >> 1) & uint32_t(kLoads - 1)) << 4 : 0u;
    const uint32_t regAddr0 = regsBase + tdfaRegLane<TdfaReg>(tid) * uint32_t(sizeof(TdfaReg));
    // (wave-uniform copies for the DMA statements: the tile's LDS address and the buffer base live in SGPRs)
    const uint32_t tileU = __builtin_amdgcn_readfirstlane(stageBase);
    auto dmaStage = [&](uint32_t stageOff) {  // the wave's 64 rows x 64 bytes at line offset stageOff -> the tile
#pragma unroll
        for (int i = 0; i < kLoads; ++i)
            if (stageOff + dmaSeg < srcSpan[i]) tdfaDmaLoad16(tileU + uint32_t(i) * 1024u, uint64_t(dataAligned), srcOff[i] + stageOff);
    };

    u32x4 in[kLoads];
    u32x4 W[kLoads];
    if constexpr (DMA) {
        dmaStage(0);
        tdfaDmaWait();
#pragma unroll
        for (int k = 0; k < kLoads; ++k) W[k] = *reinterpret_cast<LdsQuadPtr>(myRow + ((uint32_t(k) * 16) ^ mySwizzle));
    } else {
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {  // stage 0
            in[i] = u32x4{0, 0, 0, 0};
            if (seg < srcSpan[i]) in[i] = *reinterpret_cast<GlobalQuadPtr>(dataAligned + srcOff[i]);
        }
        tdfaWaveLdsSync();
#pragma unroll
        for (int i = 0; i < kLoads; ++i) *reinterpret_cast<LdsQuadPtr>(dstAddr[i]) = in[i];
        tdfaWaveLdsSync();
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {  // stage 1
            in[i] = u32x4{0, 0, 0, 0};
            if (kTdfaStageBytes + seg < srcSpan[i]) in[i] = *reinterpret_cast<GlobalQuadPtr>(dataAligned + (srcOff[i] + kTdfaStageBytes));
        }
#pragma unroll
        for (int k = 0; k < kLoads; ++k) W[k] = *reinterpret_cast<LdsQuadPtr>(myRow + ((uint32_t(k) * 16) ^ mySwizzle));
    }
    constexpr int NB = LC_TDFA_STREAM_CHUNK;  // bytes per chunk
    constexpr int kChunksPerStage = int(kTdfaStageBytes) / NB;
    // word w of the staged row: W[w / 4][w % 4]
    auto rowWord = [&](int w) -> uint32_t {
        const u32x4& q = W[(w >> 2) & (kLoads - 1)];
        return (w & 3) == 0 ? q.x : (w & 3) == 1 ? q.y : (w & 3) == 2 ? q.z : q.w;
    };
    constexpr int NC = PAIR ? NB / 2 : NB;  // chain links per chunk
    uint32_t col[NC], ptt[NC];
    {  // classes of the very first chunk (the only burst of lookups in a line's life)
        const uint32_t base0 = 0u - head;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const uint32_t b = (rowWord(j >> 2) >> ((j & 3) * 8)) & 0xFFu;
            uint32_t c;
            if constexpr ((LAB & kLabPreClass) != 0) c = (base0 + j < L) ? b : idCol;
            else if (PAIR && (j & 1) == 0) c = *reinterpret_cast<LdsHalfPtr>((base0 + j < L) ? pi.cmapA + b * 2 : idAAddr);
            else c = *reinterpret_cast<LdsBytePtr>((base0 + j < L) ? TD_CMAP_OFFSET + b : kTdfaIdColByteAddr);
            if constexpr (PAIR) {
                if ((j & 1) == 0) col[j / 2] = c;
                else col[j / 2] += c;
            } else {
                col[j] = c;
            }
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) ptt[j] = dummyT;
    }
    uint32_t pbase = 0;

    if constexpr ((LAB & kLabNoLoop) != 0) maxStages = 0;
    for (uint32_t s = 0; s < maxStages; ++s) {
        if constexpr (DMA) {
            // W holds the whole row of stage s and the tile is free: stage s+1 goes into it (needed when chunk 6 is done)
            tdfaLdsDrain();
            dmaStage((s + 1) * kTdfaStageBytes);
        } else {
            // the tile's stage s has been read by every lane (W): publish stage s+1, put stage s+2 in flight
            tdfaWaveLdsSync();
#pragma unroll
            for (int i = 0; i < kLoads; ++i) *reinterpret_cast<LdsQuadPtr>(dstAddr[i]) = in[i];
            tdfaWaveLdsSync();
            const uint32_t nextOff = (s + 2) * kTdfaStageBytes;
#pragma unroll
            for (int i = 0; i < kLoads; ++i) {
                in[i] = u32x4{0, 0, 0, 0};
                if (nextOff + seg < srcSpan[i]) in[i] = *reinterpret_cast<GlobalQuadPtr>(dataAligned + (srcOff[i] + nextOff));
            }
        }
        // Round 5 (profiles/round5_tdfa_isa_budget.md).  Whether the NEXT chunk lies wholly inside the line used to be asked per chunk
        // and lane -- six VALU and three SALU instructions per chunk, 0.75 VALU per line byte, and a branch per chunk that kept the
        // compiler from scheduling across chunks.  It is a question about the wave's largest head and smallest head + length (taken
        // once per kernel) and the chunk's position: scalar arithmetic.  A stage all of whose chunks pass runs a copy of the chunk loop
        // that does not ask (ALLFULL: straight-line code, eight chunks scheduled as one); the first stage of a line (head bytes in
        // chunks 0-1) and the last full one (only chunk 7's next chunk is short) run the asking copy for those chunks and the straight
        // copy for the rest.  Returns true when every line of the wave has ended inside [C0, C1).
        auto runChunks = [&](auto allFullTag, auto c0Tag, auto c1Tag) -> bool {
        constexpr bool ALLFULL = decltype(allFullTag)::value;
        constexpr int C0 = decltype(c0Tag)::value, C1 = decltype(c1Tag)::value;
#pragma unroll
        for (int c = C0; c < C1; ++c) {
            const uint32_t base = s * kTdfaStageBytes + uint32_t(c) * NB - head;  // line offset of byte 0 (wraps in the head)
            if constexpr (!ALLFULL && PAIR1) {
                // Every line of the wave ends before this chunk: nothing is left to walk (the stamps of the chunk before are still
                // pending and are issued behind the loop, as at a regular end).  Lines are rarely a multiple of the stage behind their
                // 16-byte row start: 512-byte lines take 8 stages and `head` (0..15) bytes of a ninth -- of whose eight chunks six or
                // seven were walked on the identity column, in the copy of the loop that tests every byte (round 5: 11 % of the
                // kernel's VALU instructions, profiles/round5_tdfa_isa_budget.md).
                if (c > 0 && __all(s * kTdfaStageBytes + uint32_t(c) * NB >= head + L)) return true;
            }
            const uint32_t nbase = base + NB;
            // Round 5: "the next chunk lies wholly inside the line, for every line of the wave" is a question about the wave's largest
            // head and smallest head + length -- two numbers taken once per kernel -- and the chunk's position: scalar arithmetic.  (It
            // was asked per chunk AND lane: six VALU instructions per chunk, 0.75 per line byte.)
            const bool waveFull = PAIR1 && waveMaxHead <= s * kTdfaStageBytes + uint32_t(c + 1) * NB &&
                                  waveMinSpan >= s * kTdfaStageBytes + uint32_t(c + 2) * NB;
            const bool fullNext = ALLFULL || waveFull || (nbase < L && L - nbase >= uint32_t(NB));
            const uint32_t entry = t;
            // the NB bytes after this chunk (the first chunk of stage s+1 after the last one of stage s: its row words were
            // re-read from the tile when chunk 0 of this stage was done)
            uint32_t nwords[NB / 4], cwords[4] = {0, 0, 0, 0};
#pragma unroll
            for (int w = 0; w < NB / 4; ++w) {
                nwords[w] = rowWord(((c + 1) * NB / 4 + w) & (kLoads * 4 - 1));
                cwords[w] = rowWord(c * NB / 4 + w);
            }
            uint32_t ncol[NC], tt[NC], seen;
            bool general;
            if constexpr (PAIR1) {
                uint32_t na[NC], nc[NC];
                constexpr bool A8 = (LAB & kLabCmapA8) != 0;
                const uint32_t mapA = A8 ? cmapA8 : pi.cmapA, idA = A8 ? cmapA8 + 256u : idAAddr;
                if (ALLFULL || waveFull || __all(fullNext)) t = tdfaStreamPair1Chunk<BLOCK, NB, false, TdfaReg, A8>(t, col, na, nc, nwords, nbase, L, mapA, idA, ptt, tt, pbase, regAddr0, reinterpret_cast<const uint8_t*>(blob) + TD_CMAP_OFFSET);
                else t = tdfaStreamPair1Chunk<BLOCK, NB, true, TdfaReg, A8>(t, col, na, nc, nwords, nbase, L, mapA, idA, ptt, tt, pbase, regAddr0);
                tdfaSettleDoubles<BLOCK, NC, TdfaReg>(ptt, pbase, regAddr0);
#pragma unroll
                for (int j = 0; j < NC; ++j) ncol[j] = A8 ? __umul24(na[j], strideA8) + nc[j] : na[j] + nc[j];
                seen = 0;
                general = false;
            } else if constexpr (PAIR) {
                uint32_t na[NC], nc[NC];
                if (__all(fullNext)) t = tdfaStreamPairChunk<BLOCK, NB, false, TdfaReg, LAB>(t, col, na, nc, nwords, nbase, L, pi.cmapA, idAAddr, ptt, tt, pbase, regAddr0, seen);
                else t = tdfaStreamPairChunk<BLOCK, NB, true, TdfaReg, LAB>(t, col, na, nc, nwords, nbase, L, pi.cmapA, idAAddr, ptt, tt, pbase, regAddr0, seen);
#pragma unroll
                for (int j = 0; j < NC; ++j) ncol[j] = na[j] + nc[j];
                general = (LAB & kLabNoGeneral) ? false : (seen & ((TP_GENERAL << 16) | (TP_GENERAL << 24))) != 0;
            } else {
                if (__all(fullNext)) t = tdfaStreamChunk<NB, false, TdfaReg, LAB>(t, col, ncol, nwords, nbase, L, idCol, ptt, tt, pbase, regAddr0, seen);
                else t = tdfaStreamChunk<NB, true, TdfaReg, LAB>(t, col, ncol, nwords, nbase, L, idCol, ptt, tt, pbase, regAddr0, seen);
                general = (LAB & kLabNoGeneral) ? false : (seen & (TD_OP_GENERAL << 16)) != 0;
            }
            if (__any(general)) {
                // (the previous chunk's stamps are all issued by now, this chunk's are not: in-order replay is exact)
                const u32x4 q = {cwords[0], cwords[1], cwords[2], cwords[3]};
                uint32_t single = entry;
                if constexpr (PAIR) single = TD_TRANS_OFFSET + ((entry & 0xFFFFu) - pi.base) / pi.rowBytes * rowBytes;
                tdfaReplayChunk<BLOCK, TdfaReg, false, LAB>(smem, q, single, base, L, idCol, regsBase, tid, NB);
#pragma unroll
                for (int j = 0; j < NC; ++j) tt[j] = dummyT;
            }
            if constexpr (DMA) {
                // the row's last segment is all chunks kChunksPerStage-2 and -1 still read: the others take stage s+1 now (the
                // last chunk's NEXT bytes are the first ones of stage s+1), the last one when the stage is done
                if (c == kChunksPerStage - 2) {
                    if constexpr ((LAB & kLabNoDmaWait) == 0) tdfaDmaWait();
#pragma unroll
                    for (int k = 0; k < kLoads - 1; ++k) W[k] = *reinterpret_cast<LdsQuadPtr>(myRow + ((uint32_t(k) * 16) ^ mySwizzle));
                } else if (c == kChunksPerStage - 1) {
                    W[kLoads - 1] = *reinterpret_cast<LdsQuadPtr>(myRow + ((uint32_t(kLoads - 1) * 16) ^ mySwizzle));
                }
            } else if ((c + 1) * NB % 16 == 0) {  // a 16-byte segment of the row is done: fetch stage s+1's
                const int k = ((c + 1) * NB / 16 - 1) & (kLoads - 1);
                W[k] = *reinterpret_cast<LdsQuadPtr>(myRow + ((uint32_t(k) * 16) ^ mySwizzle));
            }
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                col[j] = ncol[j];
                ptt[j] = tt[j];
            }
            pbase = base;
        }
        return false;
        };  // runChunks
        {
            using std::integral_constant;
            constexpr integral_constant<int, 0> k0{};
            constexpr integral_constant<int, 2> k2{};
            constexpr integral_constant<int, kChunksPerStage - 1> kLast{};
            constexpr integral_constant<int, kChunksPerStage> kEnd{};
            const uint32_t s0 = s * kTdfaStageBytes;
            // chunk c's next chunk starts at s0 + (c + 1) * NB - head (not inside the head: maxHead <= ...) and ends at s0 + (c + 2) * NB
            // - head (inside the line: minSpan >= ...)
            const bool headsPastChunk0 = PAIR1 && waveMaxHead <= s0 + uint32_t(NB);
            const bool headsPastChunk2 = PAIR1 && waveMaxHead <= s0 + 3u * uint32_t(NB);
            const bool allNextInside = PAIR1 && waveMinSpan >= s0 + kTdfaStageBytes + uint32_t(NB);
            const bool nextInsideButLast = PAIR1 && waveMinSpan >= s0 + kTdfaStageBytes;
            if (headsPastChunk0 && allNextInside) {
                (void)runChunks(std::true_type{}, k0, kEnd);
            } else if (kChunksPerStage == 8 && headsPastChunk2 && allNextInside) {
                if (!runChunks(std::false_type{}, k0, k2)) (void)runChunks(std::true_type{}, k2, kEnd);
            } else if (kChunksPerStage == 8 && headsPastChunk0 && nextInsideButLast) {
                (void)runChunks(std::true_type{}, k0, kLast);
                (void)runChunks(std::false_type{}, kLast, kEnd);
            } else {
                (void)runChunks(std::false_type{}, k0, kEnd);
            }
        }
        if (__all((t & 0xFFFFu) == deadRow || s + 1 >= myStages)) break;
    }
    if constexpr (DMA) tdfaDmaWait();  // (a stage in flight when the loop was left would land in the result tile)
    if constexpr (PAIR1) {  // the last chunk's stamps
        constexpr uint32_t kRegShift = (BLOCK == 1024 ? 12 : BLOCK == 512 ? 11 : BLOCK == 256 ? 10 : BLOCK == 128 ? 9 : 8) -
                                       (sizeof(TdfaReg) == 2 ? 1 : 0);
#pragma unroll
        for (int j = 0; j < NC; ++j)
            *reinterpret_cast<LdsRegPtr>(regAddr0 + (((ptt[j] >> 16) & 0x7Fu) << kRegShift)) = TdfaReg(pbase + 2 * j + ((ptt[j] >> 23) & 1u));
        tdfaSettleDoubles<BLOCK, NC, TdfaReg>(ptt, pbase, regAddr0);
    } else if constexpr ((LAB & kLabNoStamp) == 0) {  // the last chunk's stamps
        if constexpr (PAIR) {
            constexpr uint32_t kRegShift = (BLOCK == 1024 ? 12 : BLOCK == 512 ? 11 : BLOCK == 256 ? 10 : BLOCK == 128 ? 9 : 8) -
                                           (sizeof(TdfaReg) == 2 ? 1 : 0);
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                *reinterpret_cast<LdsRegPtr>(regAddr0 + (((ptt[j] >> 16) & 0xFFu) << kRegShift)) = TdfaReg(pbase + 2 * j);
                *reinterpret_cast<LdsRegPtr>(regAddr0 + ((ptt[j] >> 24) << kRegShift)) = TdfaReg(pbase + 2 * j + 1);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NC; ++j)
                *reinterpret_cast<LdsRegPtr>(addHighHalf(regAddr0, ptt[j])) = TdfaReg(pbase + j);
        }
    }

    if constexpr ((LAB & kLabNoOutput) != 0) {
        if (live && t == 0x12345u) status[line] = 3;  // (keeps the loop alive)
        return;
    }
    const uint32_t state = PAIR ? ((t & 0xFFFFu) - pi.base) / pi.rowBytes : ((t & 0xFFFFu) - TD_TRANS_OFFSET) / rowBytes;
    tdfaWriteResults<BLOCK, TdfaReg, LAB>(smem, stageBase, regsBase, state, live, line, L, from, order != nullptr, nGroupsOut, caps,
                                     status);
    };  // walkBlock
    if constexpr ((LAB & kLabPersist) != 0) {
        // (round 6) PERSISTENT WAVEFRONTS.  The launch has as many workgroups as the chip holds (gpu_runtime.hip); workgroup b walks
        // blocks b, b + grid, b + 2 grid, ... -- and since nothing below the table staging is shared between the wavefronts of a
        // workgroup, every wavefront goes on with ITS 64 lines of the next block the moment it has written its results: no wait for
        // the slowest of the eight before the slot is used again, tables staged once.  (Round 3 tried persistent WORKGROUPS on the
        // kernel of that time and dropped them: 0.243 against 0.227 ms.)  Between blocks the wavefront zeroes its lanes' register
        // columns ("every register starts at 0", above) behind its own LDS reads of the result tile.
        static_assert(PAIR1 && COMPACT, "persistent wavefronts: the one-stamp COMPACT kernel only");
        constexpr uint32_t kRegShift = (BLOCK == 1024 ? 12 : BLOCK == 512 ? 11 : BLOCK == 256 ? 10 : BLOCK == 128 ? 9 : 8) -
                                       (sizeof(TdfaReg) == 2 ? 1 : 0);
        const uint32_t nBlocks = (nLines + BLOCK - 1) / BLOCK;
        const uint32_t regRows = regBytes / (uint32_t(BLOCK) * uint32_t(sizeof(TdfaReg)));
        const uint32_t myRegs = regsBase + tdfaRegLane<TdfaReg>(tid) * uint32_t(sizeof(TdfaReg));
        for (uint32_t b = blockId; b < nBlocks; b += gridDim.x) {
            walkBlock(b);
            if (b + gridDim.x < nBlocks) {
                tdfaLdsDrain();
                for (uint32_t r = 0; r < regRows; ++r) *reinterpret_cast<LdsRegPtr>(myRegs + (r << kRegShift)) = TdfaReg(0);
            }
        }
    } else {
        walkBlock(blockId);
    }
}

template <int BLOCK, bool COMPACT, bool PAIR = false, int LAB = 0>
__global__ __launch_bounds__(BLOCK, (LAB & kLabWaves5) ? 5 : LC_TDFA_STREAM_WAVES) void tdfa_stream_kernel(
    const uint8_t* __restrict__ data, const uint32_t* __restrict__ off, const uint32_t* __restrict__ len, uint32_t sepBytes,
    uint32_t minLen, uint32_t nLines, const uint32_t* __restrict__ nLinesPtr, const uint32_t* __restrict__ order,
    const uint32_t* __restrict__ resume, const uint32_t* __restrict__ blob, uint32_t blobBytes, uint32_t regBytes,
    uint32_t nGroupsOut, int32_t* __restrict__ caps, uint8_t* __restrict__ status, uint32_t* __restrict__ longFlag,
    uint32_t launchSeq, uint32_t* __restrict__ doneCounter, uint32_t* __restrict__ doneFlag, uint32_t doneSeq) {
    if constexpr (!COMPACT && (LAB & kLabMopUp) != 0) {
        // (round 6) The MOP-UP launch behind a COMPACT one (minLen != 0: only the lines of 64 KiB and more are this launch's) comes with a
        // grid of at most 256 workgroups that take the line blocks in turn.  It used to come with one workgroup per block -- 4 096 for
        // the headline batch, every one of them reading the flag and leaving: 4.5 us behind each 168 us launch, on a corpus without
        // a single such line (profiles/round6_tdfa_kernel_rocprofv3.txt).  An instantiation of its own (kLabMopUp): the loop costs 12
        // VGPRs and 32 SGPRs, and the same kernel without it is what every 1000-line event group of an agent runs.
        if (minLen) {
            if (longFlag && __atomic_load_n(longFlag, __ATOMIC_RELAXED) < launchSeq) return;  // nothing was flagged by the COMPACT launch
            const uint32_t nBlocks = (nLines + BLOCK - 1) / BLOCK;
            for (uint32_t b = blockIdx.x; b < nBlocks; b += gridDim.x) {
                tdfaStreamBody<BLOCK, COMPACT, PAIR, LAB>(data, off, len, sepBytes, minLen, nLines, nLinesPtr, order, resume, blob, blobBytes,
                                                         regBytes, nGroupsOut, caps, status, longFlag, launchSeq, b);
                __syncthreads();  // (the next block's tables and tiles go where this one's were)
            }
            tdfaSignalDone(doneCounter, doneFlag, doneSeq);
            return;
        }
    }
    tdfaStreamBody<BLOCK, COMPACT, PAIR, LAB>(data, off, len, sepBytes, minLen, nLines, nLinesPtr, order, resume, blob, blobBytes,
                                             regBytes, nGroupsOut, caps, status, longFlag, launchSeq, blockIdx.x);
    tdfaSignalDone(doneCounter, doneFlag, doneSeq);
}

// ---- several batches, each with its OWN tables, in ONE launch (BASELINE configs[3]: many pipelines share the GPU; a 1000-line
// group fills 4 of 256 CUs and a launch per group leaves the chip to launch latency: measured 22 GB/s aggregate for 64
// pipelines on 4 streams against 820 GB/s for the same bytes in one launch).  Workgroup b belongs to the job whose
// [firstBlock, next firstBlock) holds b; it stages that job's tables into LDS -- the per-pipeline switch costs what staging
// 1-3 KB costs -- and walks that job's lines.
struct TdfaJob {
    const uint8_t* data;
    const uint32_t* off;
    const uint32_t* len;
    const uint32_t* blob;
    int32_t* caps;
    uint8_t* status;
    uint32_t sepBytes, nLines, blobBytes, regBytes, nGroupsOut, firstBlock;
};

// PAIR1: every job of the launch carries a ONE-STAMP byte-pair table (the launcher packs jobs with and without into separate launches)
// blockToJob[b] = the job workgroup b belongs to.  Both tables live in DEVICE memory (copied there on the launch stream): through
// round 4 every workgroup binary-searched the job table in pinned host memory -- six dependent reads across PCIe at the head of each
// of the launch's ~250 workgroups, two thirds of the launch's duration.
template <int BLOCK, bool COMPACT, bool PAIR1 = false>
__global__ __launch_bounds__(BLOCK, LC_TDFA_STREAM_WAVES) void tdfa_stream_multi_kernel(const TdfaJob* __restrict__ jobs,
                                                                                        const uint16_t* __restrict__ blockToJob,
                                                                                        uint32_t* __restrict__ doneCounter,
                                                                                        uint32_t* __restrict__ doneFlag, uint32_t doneSeq) {
    const TdfaJob j = jobs[blockToJob[blockIdx.x]];  // (wave-uniform: scalar loads)
    tdfaStreamBody<BLOCK, COMPACT, PAIR1, PAIR1 ? (kTdfaNoGeneralPrograms | kLabPairOne) : 0>(
        j.data, j.off, j.len, j.sepBytes, 0u, j.nLines, nullptr, nullptr, nullptr, j.blob, j.blobBytes, j.regBytes, j.nGroupsOut, j.caps, j.status,
        nullptr, 0u, blockIdx.x - j.firstBlock);
    tdfaSignalDone(doneCounter, doneFlag, doneSeq);
}
