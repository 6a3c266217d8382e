// tdfa_kernel.hpp -- gfx950 kernel of the tagged-DFA engine (included by gpu_runtime.hip only).
//
// One log line per lane, 64 lines per wavefront, tables in LDS.
//
// HBM side.  A lane walking its own line with its own loads touches a different cache line per lane per load, and
// with enough wavefronts in flight the lines are evicted from L2 between two visits (measured: 2.6-3.4x the
// algorithmic bytes fetched, the kernel then sits on the HBM ceiling doing wasted traffic).  So the wavefront
// loads COOPERATIVELY: per stage, four lanes fetch 64 contiguous bytes of one line (16 lines per
// global_load_dwordx4, 4 loads per stage), the 64x64-byte tile goes to a per-wave LDS staging buffer, and each lane
// then reads its own row back with ds_read_b128.  Every byte is fetched from HBM once, in 64-byte runs.  The next
// stage's loads are in flight (in VGPRs) while the current stage is stepped.  Only the wavefront itself touches
// its staging rows, so no workgroup barrier is involved.
//
// LDS side.  A stage is stepped in aligned 16-byte chunks; every chunk in three phases so that the only serial
// dependency is the DFA state itself:
//   phase 0  class lookups   col[j] = cmap8[byte j]                       16 x (v_add_sdwa + ds_read_u8), independent
//   phase 1  state chain     t = lds32[(t & 0xFFFF) + col[j]]             16 x (v_add_sdwa + ds_read_b32), dependent;
//                            the low half of a transition entry IS the LDS address of the next state's row
//   phase 2  capture writes  regs[(t_j >> 16)][lane] = pos_j              16 x (v_add_sdwa + v_add + ds_write_b32), independent;
//                            the high half of an entry IS the LDS offset of the offset register the transition
//                            stamps; transitions that stamp nothing point at a dummy register, so phase 2 has no
//                            branches and no divergence (lines hit group boundaries at different bytes: a
//                            branch per byte would be taken by some lane almost every byte)
// Transitions whose register program is more than "one register = pos" (rare for log regexes) carry a list id
// and a flag instead; a wave that saw such a flag anywhere in the chunk replays the chunk in order, with branches.
// Capture-offset registers live in LDS as regs[reg][lane]: the register number is data dependent (VGPRs would
// need scratch indexing) and [reg][lane] is bank-conflict free.
//
// Bytes outside the lane's line (head of the first aligned chunk, tail of the last, stages past the end of a
// short line) take the row's identity column.  When every lane of the wavefront is in the middle of its line a
// wave-uniform branch skips the validity selects.  Loads are aligned 16-byte segments that contain at least one
// byte of the line, so none can leave the line's pages.
//
// COMPACT variants (regex_handle.cpp packTdfaWideBlob: 256 lanes by default for large batches): 16-bit offset registers (lines of 64 KiB and more are left to a second launch
// of the 32-bit kernel) and an unpadded, XOR-swizzled staging tile -- 106 instead of 164 bytes of LDS per line, i.e. more
// lines in flight per CU.  With BYTEROWS on top (small automata, one 1024-lane workgroup per CU sharing a 37 KiB table) the
// transition rows are indexed by the BYTE itself (256 columns + the identity column, device_tables.h): phase 0 becomes
// one VALU op per byte and the class lookup -- one of the three LDS instructions per byte -- is gone.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/lc_regex_gpu.h"
#include "device_tables.h"

#ifndef LC_TDFA_MIN_WAVES
#define LC_TDFA_MIN_WAVES 1  // waves per SIMD the register allocator must leave room for
#endif

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t __attribute__((address_space(3))) * LdsWordPtr;
typedef u32x4 __attribute__((address_space(3))) * LdsQuadPtr;
typedef const uint8_t __attribute__((address_space(3))) * LdsBytePtr;
typedef const u32x4 __attribute__((address_space(1))) * GlobalQuadPtr;

// capture offsets are 32-bit (a line may be as long as a 512 KiB read buffer); the COMPACT variants keep 16-bit ones
template <typename RegT>
using LdsRegPtrT = RegT __attribute__((address_space(3))) *;
constexpr uint32_t kTdfaWideMaxLine = 0xFFFFu;  // longest line the COMPACT variants take
#ifndef LC_TDFA_CHUNK
#define LC_TDFA_CHUNK 16  // bytes stepped per three-phase round (8 halves the live col/tt registers)
#endif

#ifndef LC_TDFA_STAGE_BYTES
#define LC_TDFA_STAGE_BYTES 64
#endif
constexpr uint32_t kTdfaStageBytes = LC_TDFA_STAGE_BYTES;  // bytes of each line staged per round (32 or 64)
constexpr int kTdfaLoads = int(kTdfaStageBytes / 16);       // loads per stage == lanes that share one line
constexpr uint32_t kTdfaRowStride = kTdfaStageBytes + 16;  // +16: rows 20 dwords apart -> conflict-free b128 reads
constexpr uint32_t kTdfaStagePerWave = 64 * kTdfaRowStride;

__device__ __forceinline__ void tdfaWaveLdsSync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Measurement variants (tools/tdfa_lab.hip instantiates them; the product instantiates LAB = 0 and, for tables that carry no
// general register program at all -- device_tables.h TD_NREGS_NO_GENERAL --, kTdfaNoGeneralPrograms).  They answer "what
// does each of the three LDS instructions per byte cost" on the real kernel instead of a model of it:
//   kLabNoStamp    phase 2 is skipped (capture offsets come out wrong: timing only)
//   kLabPreClass   the input bytes ARE column offsets already (host pre-classified copy of the corpus): no class lookup
//   kLabGlobalClass the class lookup goes to the 256-byte map in global memory (vector L1) instead of LDS
//   kLabReplicated the transition table is stored 16 times, entry e of replica r at ((e * 16) + r) * 4: lane l reads replica
//                  l & 15, so the 32 lanes of an LDS lane group touch 16 banks at most twice -- no chain bank conflicts
//   kLabNoOutput / kLabNoLoop (tdfa_stream_kernel only): skip the capture-table write / the stage loop -- what a workgroup's
//                  fixed costs are (line table reads, first loads, table staging, result write)
//   kLabNoGeneral  (tdfa_stream_kernel only): no check for general register programs (wrong for tables that have one on the
//                  walked path): what the check costs -- and what tables without such programs need not pay
enum { kLabNoStamp = 1, kLabPreClass = 2, kLabGlobalClass = 4, kLabReplicated = 8, kLabNoOutput = 16, kLabNoLoop = 32, kLabNoGeneral = 64,
       kLabOneStamp = 128 /* byte-pair chunks only: one stamp per pair (timing only) */,
       kLabDmaStage = 256 /* tdfa_stream_kernel, COMPACT: the staging tile is filled by global_load_lds_dwordx4 (no staging VGPRs) */,
       kLabWaves5 = 512 /* tdfa_stream_kernel: register budget of 5 waves per SIMD (96 VGPRs) */,
       kLabNoDmaWait = 1024 /* DMA staging without the wait for the stage (wrong bytes: timing only -- what the wave loses there) */,
       kLabCmapA8 = 2048 /* one-stamp pair kernel: the first byte's class from a u8 copy of cmapA built in LDS (class INDEX, scaled on the
                            VALU): 128 ASCII bytes = 32 dwords = 32 banks, where the u16 table puts byte b and b+64 on one bank (round 6) */,
       kLabPairOne = 4096 /* byte-pair chunks on a ONE-STAMP pair table (device_tables.h TP1_*, LC_TDFA_PAIR=2): exact */,
       kLabMopUp = 8192 /* tdfa_stream_kernel: the launch behind a COMPACT one -- a small grid whose workgroups take the line blocks in turn */,
       kLabPersist = 16384 /* tdfa_stream_kernel, one-stamp pair tables: the launch is sized to the workgroups the chip holds, and every
                              WAVEFRONT goes on with the lines of its place in the next block (tables staged once, no wait for the slowest
                              wave of a workgroup before the slot is used again) */ };
constexpr uint32_t kTdfaCmapA8Bytes = 272;  // 256 class indices + the identity class's index (a byte outside the line) + padding
constexpr int kTdfaNoGeneralPrograms = kLabNoGeneral;  // the product's second instantiation (gpu_runtime.hip launchTdfaBlock)

// general register program (a list of moves); rare for log regexes
// Column of a lane in the register file.  32-bit registers: the lane itself (a wavefront's lanes hit 32 banks twice over, one
// half-wave after the other).  16-bit registers: lanes l and l+32 share a dword -- two NEIGHBOURING lanes in one dword stamp
// different registers most of the time, i.e. different dwords of one bank in the same LDS pass: a conflict on nearly every
// stamp (measured: stamps + class lookups add 1.3 conflict cycles per byte-step).
template <typename TdfaReg>
__device__ __forceinline__ uint32_t tdfaRegLane(uint32_t tid) {
    if constexpr (sizeof(TdfaReg) == 2) return (tid & ~63u) | ((tid & 31u) << 1) | ((tid >> 5) & 1u);
    else return tid;
}

template <int BLOCK, typename TdfaReg>
__device__ __forceinline__ void tdfaRunMoveList(uint8_t* smem, uint32_t regsBase, uint32_t list, uint32_t pos,
                                                uint32_t tid) {
    TdfaReg* regs = reinterpret_cast<TdfaReg*>(smem + regsBase);
    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(smem);
    const uint32_t* opsStart = reinterpret_cast<const uint32_t*>(smem + hdr[TD_OFF_OPSSTART]);
    const uint16_t* ops = reinterpret_cast<const uint16_t*>(smem + hdr[TD_OFF_OPS]);
    const uint32_t at = opsStart[list];
    const uint32_t cnt = ops[at];
#pragma unroll 1
    for (uint32_t i = 0; i < cnt; ++i) {
        const uint32_t w = ops[at + 1 + i];
        const uint32_t dst = w & 0xFF, src = w >> 8;
        const TdfaReg val = (src == TD_REG_POS) ? TdfaReg(pos) : regs[src * BLOCK + tdfaRegLane<TdfaReg>(tid)];
        regs[dst * BLOCK + tdfaRegLane<TdfaReg>(tid)] = val;
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
// from the chunk's entry state and applies every register program at its own byte.  Rolled up: it is rare.
template <int BLOCK, typename TdfaReg, bool WIDE, int LAB = 0>
__device__ __forceinline__ void tdfaReplayChunk(uint8_t* smem, u32x4 q, uint32_t t, uint32_t base, uint32_t L,
                                                uint32_t idCol, uint32_t regsBase, uint32_t tid, uint32_t nBytes) {
    typedef LdsRegPtrT<TdfaReg> LdsRegPtr;
    const LdsBytePtr cmap = reinterpret_cast<LdsBytePtr>(TD_CMAP_OFFSET);
    const uint32_t regAddr0 = regsBase + tdfaRegLane<TdfaReg>(tid) * sizeof(TdfaReg);
#pragma unroll 1
    for (uint32_t j = 0; j < nBytes; ++j) {
        const uint32_t word = (j < 8) ? ((j < 4) ? q.x : q.y) : ((j < 12) ? q.z : q.w);
        const uint32_t b = (word >> ((j & 3) * 8)) & 0xFFu;
        uint32_t col = (base + j < L) ? (WIDE ? b * 4u : ((LAB & kLabPreClass) ? b : uint32_t(cmap[b]))) : idCol;
        if constexpr ((LAB & kLabReplicated) != 0) col = (col << 4) + ((tid & 15u) << 2);
        t = *reinterpret_cast<LdsWordPtr>(addLowHalf(col, t));
        if (t & (TD_OP_GENERAL << 16)) tdfaRunMoveList<BLOCK, TdfaReg>(smem, regsBase, t >> 17, base + j, tid);
        else *reinterpret_cast<LdsRegPtr>(addHighHalf(regAddr0, t)) = TdfaReg(base + j);
    }
}

// steps NB (8 or 16) consecutive bytes held in `words`; CHECKED=false is the mid-line fast path (all NB bytes belong
// to the line)
template <int BLOCK, bool CHECKED, int NB, typename TdfaReg = uint32_t, bool WIDE = false, int LAB = 0>
__device__ __forceinline__ uint32_t tdfaStepBytes(uint8_t* smem, const uint32_t (&words)[NB / 4], uint32_t t,
                                                  uint32_t base, uint32_t L, uint32_t idCol, uint32_t regsBase,
                                                  uint32_t tid, const uint8_t* __restrict__ gcmap = nullptr) {
    typedef LdsRegPtrT<TdfaReg> LdsRegPtr;
    // the blob sits at LDS address 0, so table offsets are LDS addresses
    const LdsBytePtr cmap = reinterpret_cast<LdsBytePtr>(TD_CMAP_OFFSET);
    const uint32_t regAddr0 = regsBase + tdfaRegLane<TdfaReg>(tid) * sizeof(TdfaReg);  // LDS address of regs[0][lane]
    const uint32_t entry = t;
    uint32_t col[NB], tt[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {  // phase 0
        uint32_t c4;
        if constexpr (WIDE) {  // rows are indexed by the byte itself: column offset = byte * 4, no lookup
            const uint32_t w = words[j >> 2];
            c4 = (j & 3) == 0 ? (w << 2) & 0x3FCu : (w >> ((j & 3) * 8 - 2)) & 0x3FCu;
        } else if constexpr ((LAB & kLabPreClass) != 0) {
            c4 = (words[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
        } else if constexpr ((LAB & kLabGlobalClass) != 0) {
            c4 = gcmap[(words[j >> 2] >> ((j & 3) * 8)) & 0xFFu];
        } else {
            c4 = cmap[(words[j >> 2] >> ((j & 3) * 8)) & 0xFFu];
        }
        col[j] = CHECKED ? ((base + j < L) ? c4 : idCol) : c4;
        if constexpr ((LAB & kLabReplicated) != 0) col[j] = (col[j] << 4) + ((tid & 15u) << 2);
    }
    uint32_t seen = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {  // phase 1
        t = *reinterpret_cast<LdsWordPtr>(addLowHalf(col[j], t));
        tt[j] = t;
        seen |= t;
    }
    if (!__any((seen & (TD_OP_GENERAL << 16)) != 0)) {
        if constexpr ((LAB & kLabNoStamp) == 0) {
#pragma unroll
            for (int j = 0; j < NB; ++j)  // phase 2
                *reinterpret_cast<LdsRegPtr>(addHighHalf(regAddr0, tt[j])) = TdfaReg(base + j);
        }
    } else {
        u32x4 q = {0, 0, 0, 0};
        q.x = words[0];
        q.y = words[1];
        if (NB == 16) {
            q.z = words[2 % (NB / 4)];
            q.w = words[3 % (NB / 4)];
        }
        tdfaReplayChunk<BLOCK, TdfaReg, WIDE, LAB>(smem, q, entry, base, L, idCol, regsBase, tid, NB);
    }
    return t;
}

// Tables whose multi-stamp register programs were folded into set registers (regex_handle.cpp, planTdfaFold; the fold words
// of device_tables.h TD_NREGS) read a register as the LARGEST of itself and its sets' registers: every register starts at 0.
template <int BLOCK>
__device__ __forceinline__ void tdfaClearRegisters(uint8_t* smem, const uint32_t* __restrict__ blob, uint32_t blobBytes,
                                                   uint32_t regBytes) {
    if ((blob[TD_NREGS] >> 16) & 0x1FFFu) {  // (wave-uniform: a scalar load of the header word)
        uint32_t* regs = reinterpret_cast<uint32_t*>(smem + blobBytes);
        for (uint32_t i = threadIdx.x; i < regBytes / 4; i += BLOCK) regs[i] = 0;
    }
}

// The result of a wavefront's 64 lines: status byte + 2*nGroupsOut capture offsets per line.  A lane storing its own row
// dword by dword touches 64 different cache lines per store instruction (rows are 8*G bytes apart): measured on the headline
// batch that write alone took a third of the kernel (tools/tdfa_lab.hip, "no output" variant).  When the wave's lines are
// consecutive in the capture table (no schedule permutation, every lane decides its line) the rows form ONE contiguous
// region, so they go through the wave's staging tile (free once the line is walked): lanes write their rows to LDS, then the
// wave copies the region out 256 contiguous bytes per store instruction.  `state` = final DFA state of the lane's line.
template <int BLOCK, typename TdfaReg, int LAB = 0>
__device__ __forceinline__ void tdfaWriteResults(uint8_t* smem, uint32_t tileAddr, uint32_t regsBase, uint32_t state, bool live,
                                                 uint32_t line, uint32_t L, uint32_t from, bool permuted, uint32_t nGroupsOut,
                                                 int32_t* __restrict__ caps, uint8_t* __restrict__ status) {
    // Round 3: this epilogue was a chain of ~100 dependent LDS round trips per wave (a map byte, then a register, per slot and
    // pass; a dword per lane and store in the copy) -- 9 % of the kernel with nothing to overlap it when the workgroups of a CU
    // run in lockstep.  Now: the wave-uniform words (fold words, slot map) are read ONCE, one per lane, and handed round with
    // v_readlane; registers are read eight at a time; the tile leaves in 16-byte pieces.
    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(smem);
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t nSlots = hdr[TD_NSLOTS];
    const uint16_t* finalId = reinterpret_cast<const uint16_t*>(smem + hdr[TD_OFF_FINALID]);
    const uint8_t* finalMap = smem + hdr[TD_OFF_FINALMAP];
    const TdfaReg* regs = reinterpret_cast<const TdfaReg*>(smem + regsBase);
    const uint32_t fid = live ? uint32_t(finalId[state]) : 0xFFFFu;
    const bool matched = live && (state != 0) && (fid != 0xFFFFu);
    const uint32_t nOut = 2 * nGroupsOut;
    const uint32_t rl = tdfaRegLane<TdfaReg>(tid);
    const uint32_t foldOff = ((hdr[TD_NREGS] >> 16) & 0x1FFFu) * 16;
    if (foldOff) {
        // folded multi-stamp programs: a member register reads as the larger of itself and its set's register ("latest stamp"
        // = "largest offset", registers start at 0).  Settled here, once per line, so the row building below stays as it is.
        const uint32_t* fw = reinterpret_cast<const uint32_t*>(smem + foldOff);
        TdfaReg* rw = reinterpret_cast<TdfaReg*>(smem + regsBase) + tdfaRegLane<TdfaReg>(tid);
        const uint32_t nWords = __builtin_amdgcn_readfirstlane(fw[0]);
        for (uint32_t i0 = 0; i0 < nWords; i0 += 64) {
            const uint32_t fwv = (i0 + lane < nWords) ? fw[1 + i0 + lane] : 0xFFFFFF00u;  // (lane i holds word i0 + i)
            const uint32_t cnt = nWords - i0 < 64 ? nWords - i0 : 64;
            for (uint32_t j0 = 0; j0 < cnt; j0 += 4) {
                uint32_t w[4];
                TdfaReg both[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {  // four set registers per round trip
                    w[k] = uint32_t(__builtin_amdgcn_readlane(int(fwv), int((j0 + k) & 63)));
                    both[k] = rw[(w[k] & 0xFFu) * BLOCK];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (j0 + k >= cnt) break;
                    if (!__any(both[k] != 0)) continue;  // no line of the wave took such a transition (the usual case)
                    // (a set register is never the member of another set: the four reads above stay valid)
#pragma unroll
                    for (int m = 1; m < 4; ++m) {
                        const uint32_t r = (w[k] >> (8 * m)) & 0xFFu;
                        if (r != 0xFFu) {
                            const TdfaReg own = rw[r * BLOCK];
                            rw[r * BLOCK] = both[k] > own ? both[k] : own;
                        }
                    }
                }
            }
        }
    }
    if constexpr ((LAB & kLabPairOne) != 0) {
        // one-stamp pair tables: a register that is only ever stamped one byte behind another one lost its own stamps and reads
        // as that one + delta (device_tables.h TP_OFF_DERIVE; in table order: every source is settled before it is used)
        const uint32_t* ph = reinterpret_cast<const uint32_t*>(smem + hdr[TD_OFF_PAIR]);
        const uint32_t dOff = ph[TP_OFF_DERIVE];
        if (dOff) {
            const uint32_t* dw = reinterpret_cast<const uint32_t*>(smem + dOff);
            TdfaReg* rw = reinterpret_cast<TdfaReg*>(smem + regsBase) + tdfaRegLane<TdfaReg>(tid);
            const uint32_t nWords = __builtin_amdgcn_readfirstlane(dw[0]);
            for (uint32_t i = 0; i < nWords; ++i) {
                const uint32_t w = __builtin_amdgcn_readfirstlane(dw[1 + i]);
                rw[(w & 0xFFu) * BLOCK] = TdfaReg(rw[((w >> 8) & 0xFFu) * BLOCK] + TdfaReg(w >> 16));
            }
        }
    }
    auto slotValue = [&](uint32_t s) -> int32_t {
        int32_t val = -1;
        if (matched && s < nSlots) {
            const uint32_t m = finalMap[fid * nSlots + s];
            if (m == TD_REG_POS) val = int32_t(L + from);
            else if (m != TD_REG_NONE) val = int32_t(regs[m * BLOCK + rl] + from);
        }
        return val;
    };
    constexpr uint32_t kTileWords = 1024;  // 64 rows x 64 bytes: the smaller of the two tile layouts
    if (!permuted && nOut != 0 && nOut <= kTileWords && __all(live)) {
        int32_t* tile = reinterpret_cast<int32_t*>(smem + tileAddr);
        uint32_t rowsPerPass = kTileWords / nOut;
        if (rowsPerPass > 64) rowsPerPass = 64;
        if (rowsPerPass > 1) rowsPerPass &= ~1u;  // even: every pass starts on a 16-byte boundary of the capture table (nOut is even)
        int32_t* gout = caps + size_t(line - lane) * nOut;  // (lines of the wave are consecutive: line - lane = lane 0's)
        // Lines of one format end in the same accepting state: then the slot -> register map is the same for the whole wave
        // and the row is built without a per-slot, per-lane map lookup and its branches.
        const uint32_t fid0 = __builtin_amdgcn_readfirstlane(fid);
        const bool sameMap = __all(fid == fid0) && fid0 != 0xFFFFu;
        const int32_t end = int32_t(L + from);
        for (uint32_t p0 = 0; p0 < 64; p0 += rowsPerPass) {
            tdfaWaveLdsSync();
            const bool inPass = lane >= p0 && lane - p0 < rowsPerPass;
            int32_t* row = tile + (inPass ? lane - p0 : 0u) * nOut;
            if (sameMap) {
                const uint8_t* map = finalMap + fid0 * nSlots;
                for (uint32_t s0 = 0; s0 < nOut; s0 += 64) {  // (one trip unless the pattern has more than 32 groups)
                    const uint32_t sl = s0 + lane;
                    const uint32_t mapv = (sl < nSlots && sl < nOut) ? uint32_t(map[sl]) : uint32_t(TD_REG_NONE);  // lane i: slot s0 + i
                    const uint32_t cnt = nOut - s0 < 64 ? nOut - s0 : 64;
                    for (uint32_t k0 = 0; k0 < cnt; k0 += 8) {
                        int32_t v[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {  // (m is wave-uniform; the register read is issued whatever m says)
                            const uint32_t m = uint32_t(__builtin_amdgcn_readlane(int(mapv), int((k0 + k) & 63)));
                            const uint32_t r = m < TD_REG_NONE ? m : 0u;
                            const int32_t reg = int32_t(regs[r * BLOCK + rl] + from);
                            int32_t val = m == TD_REG_POS ? end : reg;
                            val = m == TD_REG_NONE ? -1 : val;
                            v[k] = state != 0 ? val : -1;
                        }
                        if (inPass) {
#pragma unroll
                            for (int k = 0; k < 8; ++k)
                                if (k0 + k < cnt) row[s0 + k0 + k] = v[k];
                        }
                    }
                }
            } else if (inPass) {
#pragma unroll 4
                for (uint32_t s = 0; s < nOut; ++s) row[s] = slotValue(s);
            }
            tdfaWaveLdsSync();
            const uint32_t rows = 64 - p0 < rowsPerPass ? 64 - p0 : rowsPerPass;
            const uint32_t total = rows * nOut;
            int32_t* g = gout + size_t(p0) * nOut;
            if ((reinterpret_cast<uintptr_t>(g) & 15u) == 0 && (total & 3u) == 0) {  // (wave-uniform)
                typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
                const uint32_t nq = total / 4;  // <= 256: at most four 16-byte pieces per lane
                i32x4 q[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t at = lane + 64u * uint32_t(i);
                    if (at < nq) q[i] = *reinterpret_cast<const i32x4 __attribute__((address_space(3)))*>(tileAddr + at * 16u);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t at = lane + 64u * uint32_t(i);
                    // (non-temporal: the capture table is written once and not read again by this kernel; round 3 lab: -1 %)
                    if (at < nq) __builtin_nontemporal_store(q[i], reinterpret_cast<i32x4*>(g) + at);
                }
            } else {
                for (uint32_t d = lane; d < total; d += 64) g[d] = tile[d];
            }
        }
    } else if (live) {
        int32_t* out = caps + size_t(line) * nOut;
        for (uint32_t s = 0; s < nOut; ++s) out[s] = slotValue(s);
    }
    if (live) status[line] = matched ? LC_MATCH : LC_NOMATCH;
}

// Completion signal of a launch whose caller polls host memory instead of asking the runtime (gpu_runtime.hip, the zero-copy
// host path): every workgroup publishes its results system-wide and counts itself; the last one resets the counter and stores
// the launch's sequence number into the caller's pinned flag word.  doneFlag == nullptr: nothing to do.
__device__ __forceinline__ void tdfaSignalDone(uint32_t* doneCounter, uint32_t* doneFlag, uint32_t doneSeq) {
    if (!doneFlag) return;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t prev = __hip_atomic_fetch_add(doneCounter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {
            __hip_atomic_store(doneCounter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(doneFlag, doneSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Byte-PAIR stepping (device_tables.h TP_*): the state chain -- the only serial dependency, one LDS round trip per link --
// has one link per TWO bytes: t = pair32[(t & 0xFFFF) + cmapA[byte 2k] + cmap8[byte 2k+1]].  The entry also names the
// register each of the two bytes stamps.  Same phases as tdfaStepBytes; NB/2 dependent lookups instead of NB.
struct TdfaPairInfo {
    uint32_t base, rowBytes, cmapA, idA;
};
typedef const uint16_t __attribute__((address_space(3))) * LdsHalfPtr;

template <int BLOCK, bool CHECKED, int NB, typename TdfaReg = uint32_t>
__device__ __forceinline__ uint32_t tdfaStepPairs(uint8_t* smem, const uint32_t (&words)[NB / 4], uint32_t t,
                                                  uint32_t base, uint32_t L, uint32_t idCol, uint32_t regsBase,
                                                  uint32_t tid, const TdfaPairInfo& pi, uint32_t singleRowBytes) {
    typedef LdsRegPtrT<TdfaReg> LdsRegPtr;
    const LdsBytePtr cmap = reinterpret_cast<LdsBytePtr>(TD_CMAP_OFFSET);
    const uint32_t regAddr0 = regsBase + tdfaRegLane<TdfaReg>(tid) * sizeof(TdfaReg);
    static_assert((BLOCK & (BLOCK - 1)) == 0 && BLOCK >= 64 && BLOCK <= 1024, "register stride must be a power of two");
    constexpr uint32_t kRegShift = (BLOCK == 1024 ? 12 : BLOCK == 512 ? 11 : BLOCK == 256 ? 10 : BLOCK == 128 ? 9 : 8) -
                                   (sizeof(TdfaReg) == 2 ? 1 : 0);  // log2(BLOCK * sizeof(TdfaReg))
    const uint32_t entry = t;
    uint32_t col[NB / 2], tt[NB / 2];
#pragma unroll
    for (int p = 0; p < NB / 2; ++p) {  // phase 0: two class lookups per pair, independent
        const int j = 2 * p;
        const uint32_t b0 = (words[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
        const uint32_t b1 = (words[(j + 1) >> 2] >> (((j + 1) & 3) * 8)) & 0xFFu;
        uint32_t a = *reinterpret_cast<LdsHalfPtr>(pi.cmapA + b0 * 2);
        uint32_t c = cmap[b1];
        if (CHECKED) {
            a = (base + j < L) ? a : pi.idA;
            c = (base + j + 1 < L) ? c : idCol;
        }
        col[p] = a + c;
    }
    uint32_t seen = 0;
#pragma unroll
    for (int p = 0; p < NB / 2; ++p) {  // phase 1: the state chain, one link per pair
        t = *reinterpret_cast<LdsWordPtr>(addLowHalf(col[p], t));
        tt[p] = t;
        seen |= t;
    }
    if (!__any((seen & ((TP_GENERAL << 16) | (TP_GENERAL << 24))) != 0)) {
#pragma unroll
        for (int p = 0; p < NB / 2; ++p) {  // phase 2: capture writes, two per pair, independent
            const uint32_t r0 = (tt[p] >> 16) & 0xFFu, r1 = tt[p] >> 24;
            *reinterpret_cast<LdsRegPtr>(regAddr0 + (r0 << kRegShift)) = TdfaReg(base + 2 * p);
            *reinterpret_cast<LdsRegPtr>(regAddr0 + (r1 << kRegShift)) = TdfaReg(base + 2 * p + 1);
        }
    } else {  // a general register program somewhere in the chunk: replay it byte by byte on the single-byte table
        u32x4 q = {0, 0, 0, 0};
        q.x = words[0];
        q.y = words[1];
        if (NB == 16) {
            q.z = words[2 % (NB / 4)];
            q.w = words[3 % (NB / 4)];
        }
        const uint32_t state = ((entry & 0xFFFFu) - pi.base) / pi.rowBytes;
        tdfaReplayChunk<BLOCK, TdfaReg, false>(smem, q, TD_TRANS_OFFSET + state * singleRowBytes, base, L, idCol, regsBase, tid,
                                               NB);
    }
    return t;
}

// minLen: lines shorter than this are not this launch's business (the 32-bit kernel mopping up behind a COMPACT one)
template <int BLOCK, bool PAIR, bool COMPACT = false, bool BYTEROWS = false, int LAB = 0>
__global__ __launch_bounds__(BLOCK, LC_TDFA_MIN_WAVES) void tdfa_match_kernel(const uint8_t* __restrict__ data,
                                                           const uint32_t* __restrict__ off,
                                                           const uint32_t* __restrict__ len, uint32_t sepBytes,
                                                           uint32_t minLen,
                                                           uint32_t nLines, const uint32_t* __restrict__ nLinesPtr,
                                                           const uint32_t* __restrict__ order,
                                                           const uint32_t* __restrict__ resume,
                                                           const uint32_t* __restrict__ blob,
                                                           uint32_t blobBytes, uint32_t regBytes, uint32_t nGroupsOut,
                                                           int32_t* __restrict__ caps, uint8_t* __restrict__ status,
                                                           uint32_t* __restrict__ longFlag, uint32_t launchSeq,
                                                           uint32_t* __restrict__ doneCounter, uint32_t* __restrict__ doneFlag,
                                                           uint32_t doneSeq) {
    static_assert(!(BYTEROWS && PAIR) && (COMPACT || !BYTEROWS), "byte rows: compact only, and no pair extension");
    constexpr bool WIDE = BYTEROWS;
    typedef typename std::conditional<COMPACT, uint16_t, uint32_t>::type TdfaReg;
    // staging rows: padded to 80 bytes (conflict-free b128 reads), or -- COMPACT -- 64 bytes with the 16-byte segments of
    // row r stored at segment ^ ((r >> 1) & 3), which is conflict-free without the padding
    constexpr uint32_t kRowStride = COMPACT ? kTdfaStageBytes : kTdfaRowStride;
    constexpr uint32_t kStagePerWave = 64 * kRowStride;
    static_assert(!COMPACT || kTdfaStageBytes == 64, "the swizzle is written for 4 segments per row");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    if (nLinesPtr) {  // line count produced on the device (split kernels) -- no host round trip between the launches
        const uint32_t dyn = *nLinesPtr;
        nLines = dyn < nLines ? dyn : nLines;
    }
    // The COMPACT launch raises *longFlag to its sequence number when it meets a line it has to leave behind; the mop-up
    // launch that follows it on the stream has nothing to do while the flag is older than that.
    if (minLen && longFlag && __atomic_load_n(longFlag, __ATOMIC_RELAXED) < launchSeq) return;
    if (minLen) {  // mop-up launch: usually no line of this workgroup is long enough -- leave before staging the tables
        const uint32_t s0 = blockIdx.x * BLOCK + tid;
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
        // (not __syncthreads_or: its workgroup reduction brings static LDS of its own, and the tables must sit at LDS
        // address 0)
        volatile uint32_t* flag = reinterpret_cast<volatile uint32_t*>(smem);
        if (tid == 0) *flag = 0;
        __syncthreads();
        if (mine) *flag = 1;
        __syncthreads();
        const bool any = *flag != 0;
        __syncthreads();  // everybody has read the flag before the tables overwrite it
        if (!any) return;
    }
    {  // stage the tables: 16-byte coalesced copies
        const uint4* src = reinterpret_cast<const uint4*>(blob);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = tid; i < blobBytes / 16; i += BLOCK) dst[i] = src[i];
        tdfaClearRegisters<BLOCK>(smem, blob, blobBytes, regBytes);
    }
    __syncthreads();
    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(smem);
    const uint32_t rowBytes = hdr[TD_ROW_BYTES];
    const uint32_t idCol = hdr[TD_ID_COL];
    const uint32_t regsBase = blobBytes;
    uint32_t t = hdr[TD_START_ROW];  // low 16 bits: LDS address of the current state's row
    TdfaPairInfo pi{};
    // single-byte row address -> the row the kernel actually walks (PAIR: the same state's row in the pair table)
    auto walkRow = [&](uint32_t singleRow) {
        return PAIR ? pi.base + ((singleRow & 0xFFFFu) - TD_TRANS_OFFSET) / rowBytes * pi.rowBytes : singleRow;
    };
    if constexpr (PAIR) {
        const uint32_t* ph = reinterpret_cast<const uint32_t*>(smem + hdr[TD_OFF_PAIR]);
        pi = TdfaPairInfo{ph[TP_BASE], ph[TP_ROW_BYTES], ph[TP_OFF_CMAPA], ph[TP_ID_A]};
        t = walkRow(t);
    }
    const uint32_t deadRow = PAIR ? pi.base : TD_TRANS_OFFSET;
    const uint8_t* gcmap = reinterpret_cast<const uint8_t*>(blob) + TD_CMAP_OFFSET;  // (LAB variants only)

    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t stageBase = blobBytes + regBytes + wave * kStagePerWave;  // this wave's staging rows (LDS address)

    const uint32_t slot = blockIdx.x * BLOCK + tid;
    bool live = slot < nLines;
    const uint32_t line = (live && order) ? order[slot] : slot;  // length-aware schedule (sched_kernel.hpp)
    uint32_t o = 0, L = 0;
    uint32_t from = 0;  // search patterns: offset inside the line where this search resumes (0 = a fresh search)
    if (live) {
        o = off[line];
        L = len ? len[line] : off[line + 1] - o - sepBytes;
        if (resume) {
            from = resume[line];
            from = from < L ? from : L;
            if (from) {  // only the wrapper's prefix thread is alive; what it remembers is the class of the previous byte
                const uint32_t* startAfter = reinterpret_cast<const uint32_t*>(smem + hdr[TD_OFF_STARTAFTER]);
                t = walkRow(startAfter[smem[TD_CMAP_OFFSET + data[size_t(o) + from - 1]] >> 2]);
                o += from;
                L -= from;
            }
        }
        if ((COMPACT && L > kTdfaWideMaxLine) || L < minLen) {  // another launch decides this line
            if (COMPACT && L > kTdfaWideMaxLine && longFlag) atomicMax(longFlag, launchSeq);
            live = false;
            L = 0;
        }
    }
    const uintptr_t addr = reinterpret_cast<uintptr_t>(data) + o;
    const uint32_t head = uint32_t(addr & 15);
    const uintptr_t rowStart = addr - head;              // 16-byte aligned start of this lane's line
    const uint32_t span = L ? head + L : 0;              // staged bytes [0, span) of the aligned run are needed
    const uint32_t myStages = (span + kTdfaStageBytes - 1) / kTdfaStageBytes;
    uint32_t maxStages = myStages;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t other = __shfl_xor(maxStages, d, 64);
        maxStages = other > maxStages ? other : maxStages;
    }

    // cooperative load plan: kTdfaLoads lanes share one line; load i of a stage serves line r = (64/kTdfaLoads)*i +
    // lane/kTdfaLoads, 16-byte segment lane%kTdfaLoads
    const uint32_t seg = (lane % kTdfaLoads) * 16;
    uintptr_t srcAddr[kTdfaLoads];
    uint32_t srcSpan[kTdfaLoads], dstAddr[kTdfaLoads];
#pragma unroll
    for (int i = 0; i < kTdfaLoads; ++i) {
        const int r = (64 / kTdfaLoads) * i + int(lane / kTdfaLoads);
        const uint32_t lo = __shfl(uint32_t(rowStart), r, 64);
        const uint32_t hi = __shfl(uint32_t(rowStart >> 32), r, 64);
        srcAddr[i] = ((uintptr_t(hi) << 32) | lo) + seg;
        srcSpan[i] = __shfl(span, r, 64);
        dstAddr[i] = stageBase + uint32_t(r) * kRowStride + (COMPACT ? seg ^ (((uint32_t(r) >> 1) & 3u) << 4) : seg);
    }
    const uint32_t myRow = stageBase + lane * kRowStride;
    const uint32_t mySwizzle = COMPACT ? ((lane >> 1) & 3u) << 4 : 0u;

    u32x4 in[kTdfaLoads];
#pragma unroll
    for (int i = 0; i < kTdfaLoads; ++i) {  // stage 0
        in[i] = u32x4{0, 0, 0, 0};
        if (seg < srcSpan[i]) in[i] = *reinterpret_cast<GlobalQuadPtr>(srcAddr[i]);
    }

    for (uint32_t s = 0; s < maxStages; ++s) {
        // publish stage s to LDS, then put stage s+1 in flight
        tdfaWaveLdsSync();
#pragma unroll
        for (int i = 0; i < kTdfaLoads; ++i) *reinterpret_cast<LdsQuadPtr>(dstAddr[i]) = in[i];
        tdfaWaveLdsSync();
        const uint32_t nextOff = (s + 1) * kTdfaStageBytes;
#pragma unroll
        for (int i = 0; i < kTdfaLoads; ++i) {
            in[i] = u32x4{0, 0, 0, 0};
            if (nextOff + seg < srcSpan[i]) in[i] = *reinterpret_cast<GlobalQuadPtr>(srcAddr[i] + nextOff);
        }
#pragma unroll 1
        for (uint32_t k = 0; k < kTdfaStageBytes / 16; ++k) {
            const u32x4 q = *reinterpret_cast<LdsQuadPtr>(myRow + ((k * 16) ^ mySwizzle));
            const uint32_t base = s * kTdfaStageBytes + k * 16 - head;  // line offset of byte 0 (wraps in the head)
            const bool full = base < L && L - base >= 16;
#if LC_TDFA_CHUNK == 16
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
            if constexpr (PAIR) {
                if (__all(full)) t = tdfaStepPairs<BLOCK, false, 16, TdfaReg>(smem, w, t, base, L, idCol, regsBase, tid, pi, rowBytes);
                else t = tdfaStepPairs<BLOCK, true, 16, TdfaReg>(smem, w, t, base, L, idCol, regsBase, tid, pi, rowBytes);
            } else {
                if (__all(full)) t = tdfaStepBytes<BLOCK, false, 16, TdfaReg, WIDE, LAB>(smem, w, t, base, L, idCol, regsBase, tid, gcmap);
                else t = tdfaStepBytes<BLOCK, true, 16, TdfaReg, WIDE, LAB>(smem, w, t, base, L, idCol, regsBase, tid, gcmap);
            }
#else
            const uint32_t w0[2] = {q.x, q.y}, w1[2] = {q.z, q.w};
            if constexpr (PAIR) {
                if (__all(full)) {
                    t = tdfaStepPairs<BLOCK, false, 8, TdfaReg>(smem, w0, t, base, L, idCol, regsBase, tid, pi, rowBytes);
                    t = tdfaStepPairs<BLOCK, false, 8, TdfaReg>(smem, w1, t, base + 8, L, idCol, regsBase, tid, pi, rowBytes);
                } else {
                    t = tdfaStepPairs<BLOCK, true, 8, TdfaReg>(smem, w0, t, base, L, idCol, regsBase, tid, pi, rowBytes);
                    t = tdfaStepPairs<BLOCK, true, 8, TdfaReg>(smem, w1, t, base + 8, L, idCol, regsBase, tid, pi, rowBytes);
                }
            } else if (__all(full)) {
                t = tdfaStepBytes<BLOCK, false, 8, TdfaReg, WIDE>(smem, w0, t, base, L, idCol, regsBase, tid);
                t = tdfaStepBytes<BLOCK, false, 8, TdfaReg, WIDE>(smem, w1, t, base + 8, L, idCol, regsBase, tid);
            } else {
                t = tdfaStepBytes<BLOCK, true, 8, TdfaReg, WIDE>(smem, w0, t, base, L, idCol, regsBase, tid);
                t = tdfaStepBytes<BLOCK, true, 8, TdfaReg, WIDE>(smem, w1, t, base + 8, L, idCol, regsBase, tid);
            }
#endif
        }
        // every lane dead (or past its end in the identity column): nothing left to decide for this wavefront
        if (__all((t & 0xFFFFu) == deadRow || s + 1 >= myStages)) break;
    }

    const uint32_t state = PAIR ? ((t & 0xFFFFu) - pi.base) / pi.rowBytes : ((t & 0xFFFFu) - TD_TRANS_OFFSET) / rowBytes;
    tdfaWriteResults<BLOCK, TdfaReg>(smem, stageBase, regsBase, state, live, line, L, from, order != nullptr, nGroupsOut, caps,
                                     status);
    tdfaSignalDone(doneCounter, doneFlag, doneSeq);
}
