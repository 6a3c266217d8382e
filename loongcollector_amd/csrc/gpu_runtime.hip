// gpu_runtime.hip -- gfx950 kernels and the device half of the C ABI (include/lc_regex_gpu.h).
//
// Kernels (hand-written HIP for CDNA4, wave64):
//   tdfa_match_kernel : one log line per lane.  The pattern's tagged-DFA tables are staged once per workgroup
//                       into LDS; every lane walks its own line with aligned 16-byte global loads, one
//                       class lookup + one transition lookup (both LDS) per byte, and the few capture-offset
//                       register moves attached to a transition.  Registers live in LDS as [reg][lane] so that
//                       data-dependent register numbers never spill to scratch and never bank-conflict.
//   nfa_match_kernel  : one log line per wavefront, one lane per live NFA thread (priority order == lane order);
//                       follow lists in LDS, ballot/mbcnt compaction, ds_bpermute capture transfer.
// Byte-scan work: no MFMA.  The bound that matters is LDS lookup throughput / latency, then HBM.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "device_tables.h"
#include "regex_handle.hpp"

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string tlsError;
static int hipFail(hipError_t e, const char* what) {
    tlsError = std::string(what) + ": " + hipGetErrorString(e);
    return LC_ERR_HIP;
}
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e_ = (expr);                         \
        if (e_ != hipSuccess) return hipFail(e_, #expr); \
    } while (0)

extern "C" const char* lc_last_error(void) { return tlsError.c_str(); }

extern "C" int lc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// ------------------------------------------------------------------------------------------------ TDFA kernel
// One log line per lane.  Per byte: one class lookup (off the dependency chain, issued 16 at a time) and one
// transition lookup (the chain), both LDS.  Bytes outside the lane's line (alignment head / tail of the last
// 16-byte chunk) take the row's identity column, so the byte loop has no validity branch.  Capture offsets live
// in LDS as regs[reg][lane]; the register program of a transition is almost always "regs[d] = pos" and is then
// encoded in the transition word itself (device_tables.h).
struct TdfaView {  // LDS byte offsets, wave-uniform
    uint32_t cmap, trans, finalId, finalMap, opsStart, ops, regs;
};

template <int BLOCK>
__device__ __forceinline__ void tdfaRegisterProgram(uint8_t* smem, const TdfaView& v, uint32_t h, uint32_t pos,
                                                    uint32_t tid) {
    uint32_t* regs = reinterpret_cast<uint32_t*>(smem + v.regs);
    if (h & TD_OP_INLINE) {
        const uint32_t dst = h & 0xFFu;
        regs[dst * BLOCK + tid] = pos;
        if (h & TD_OP_PAIR) regs[(dst + 1) * BLOCK + tid] = pos;
        return;
    }
    const uint32_t* opsStart = reinterpret_cast<const uint32_t*>(smem + v.opsStart);
    const uint16_t* ops = reinterpret_cast<const uint16_t*>(smem + v.ops);
    const uint32_t at = opsStart[h];
    const uint32_t cnt = ops[at];
#pragma unroll 1
    for (uint32_t i = 0; i < cnt; ++i) {
        const uint32_t w = ops[at + 1 + i];
        const uint32_t dst = w & 0xFF, src = w >> 8;
        const uint32_t val = (src == TD_REG_POS) ? pos : regs[src * BLOCK + tid];
        regs[dst * BLOCK + tid] = val;
    }
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void tdfa_match_kernel(const uint8_t* __restrict__ data,
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
    TdfaView v;
    v.cmap = hdr[TD_OFF_CLASSMAP];
    v.trans = hdr[TD_OFF_TRANS];
    v.finalId = hdr[TD_OFF_FINALID];
    v.finalMap = hdr[TD_OFF_FINALMAP];
    v.opsStart = hdr[TD_OFF_OPSSTART];
    v.ops = hdr[TD_OFF_OPS];
    v.regs = blobBytes;
    const uint32_t nSlots = hdr[TD_NSLOTS];
    const uint32_t rowBytes = hdr[TD_ROW_BYTES];
    const uint32_t idCol = hdr[TD_ID_COL];
    uint32_t row = hdr[TD_START_ROW];

    const uint32_t line = blockIdx.x * BLOCK + tid;
    const bool live = line < nLines;
    uint32_t o = 0, L = 0;
    if (live) {
        o = off[line];
        L = len ? len[line] : off[line + 1] - o - sepBytes;
    }
    const uint16_t* cmap = reinterpret_cast<const uint16_t*>(smem + v.cmap);
    const uint8_t* transBase = smem + v.trans;

    // The line is walked in 64-byte windows of four aligned 16-byte loads; the next window is in flight while
    // the current one is stepped.  An aligned 16-byte chunk that holds at least one byte of the line never leaves
    // the line's pages, so no load can fault; chunks wholly outside the line are not loaded.
    const uintptr_t addr = reinterpret_cast<uintptr_t>(data) + o;
    const uint32_t head = uint32_t(addr & 15);
    const uint4* chunk = reinterpret_cast<const uint4*>(addr - head);
    const uint32_t nChunks = L ? (head + L + 15) / 16 : 0;
    const uint32_t nWindows = (nChunks + 3) / 4;
    uint4 cur[4], nxt[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        cur[q] = make_uint4(0, 0, 0, 0);
        if (uint32_t(q) < nChunks) cur[q] = chunk[q];
    }

    for (uint32_t w = 0; w < nWindows; ++w) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            nxt[q] = make_uint4(0, 0, 0, 0);
            const uint32_t c = (w + 1) * 4 + q;
            if (c < nChunks) nxt[q] = chunk[c];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t base = (w * 4 + q) * 16 - head;  // line offset of byte 0 of this chunk (wraps in the head)
            const uint32_t words[4] = {cur[q].x, cur[q].y, cur[q].z, cur[q].w};
            uint32_t col[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {  // class lookups: independent of the DFA state, issued together
                const uint32_t b = (words[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
                const uint32_t c4 = cmap[b];
                col[j] = (base + j < L) ? c4 : idCol;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {  // the dependent chain: one LDS lookup per byte
                const uint32_t t = *reinterpret_cast<const uint32_t*>(transBase + row + col[j]);
                const uint32_t h = t >> TD_LIST_SHIFT;
                if (h) tdfaRegisterProgram<BLOCK>(smem, v, h, base + j, tid);
                row = t & TD_ROW_MASK;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
        if (row == 0) break;  // dead state: regex_match can no longer succeed for this line
    }

    if (!live) return;
    const uint16_t* finalId = reinterpret_cast<const uint16_t*>(smem + v.finalId);
    const uint8_t* finalMap = smem + v.finalMap;
    const uint32_t* regs = reinterpret_cast<const uint32_t*>(smem + v.regs);
    const uint32_t state = row / rowBytes;
    const uint32_t fid = finalId[state];
    const bool matched = (row != 0) && (fid != 0xFFFFu);
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

// ------------------------------------------------------------------------------------------------ NFA kernel
// One line per wavefront.  Lane t holds the t-th live thread of the Pike VM in backtracking-priority order:
// (position, capture offsets in VGPRs).  Per input byte (wave-uniform, fetched with v_readlane from a 256-byte
// chunk the wave loaded with one coalesced dword per lane):
//   1. every lane walks its position's follow list (LDS) and keeps the paths whose target accepts the byte class
//      and whose assertions hold; ds_min on best[target] elects, per target, the candidate of highest priority
//      (rank = lexicographic (lane, path index));
//   2. winners are compacted with a wave prefix sum into the new thread order and scattered through LDS;
//   3. lane j pulls its captures from its source lane with ds_bpermute and stamps the tagged slots.
constexpr int kNfaBlock = 256;  // 4 wavefronts = 4 lines per workgroup, sharing one LDS copy of the tables
constexpr int kNfaWaves = kNfaBlock / 64;

// LDS hand-off between lanes of ONE wavefront: order the wave's own DS operations and stop the compiler from
// moving loads/stores across the hand-off (no s_barrier needed, the wave is the only party).
__device__ __forceinline__ void waveLdsSync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t waveExclusiveScan(uint32_t v, uint32_t lane, uint32_t& total) {
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= uint32_t(d)) incl += up;
    }
    total = __shfl(incl, 63, 64);
    return incl - v;
}

// bitmask of AssertKind values that hold between `prev` and `next` (wave-uniform inputs; -1 = edge of the line)
__device__ __forceinline__ uint32_t condsTrue(int prev, int next) {
    auto word = [](int c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; };
    auto sep = [](int c) { return c == '\n' || c == '\r' || c == '\f'; };
    const bool atStart = prev < 0, atEnd = next < 0;
    const bool crlf = prev == '\r' && next == '\n';
    const bool pw = !atStart && word(prev), nw = !atEnd && word(next);
    uint32_t m = 0;
    if (atStart || (sep(prev) && !crlf)) m |= 1u << 0;  // BolMulti
    if (atStart) m |= 1u << 1;                           // BolSingle
    if (atEnd || (sep(next) && !crlf)) m |= 1u << 2;     // EolMulti
    if (atEnd) m |= 1u << 3;                             // EolSingle
    if (pw != nw) m |= 1u << 4;                          // WordBoundary
    if (pw == nw) m |= 1u << 5;                          // NotWordBoundary
    if (!pw && nw) m |= 1u << 6;                         // WordStart
    if (pw && !nw) m |= 1u << 7;                         // WordEnd
    return m;
}

template <int NS>
__global__ __launch_bounds__(kNfaBlock) void nfa_match_kernel(const uint8_t* __restrict__ data,
                                                              const uint32_t* __restrict__ off,
                                                              const uint32_t* __restrict__ len, uint32_t sepBytes,
                                                              uint32_t nLines, const uint32_t* __restrict__ blob,
                                                              uint32_t blobBytes, uint32_t nGroupsOut,
                                                              int32_t* __restrict__ caps,
                                                              uint8_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    {
        const uint4* src = reinterpret_cast<const uint4*>(blob);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = tid; i < blobBytes / 16; i += kNfaBlock) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(smem);
    const uint32_t nPos = hdr[NF_NPOS];
    const uint32_t nSlots = hdr[NF_NSLOTS];
    const uint8_t* classMap = smem + hdr[NF_OFF_CLASSMAP];
    const uint2* posMask = reinterpret_cast<const uint2*>(smem + hdr[NF_OFF_POSMASK]);
    const uint32_t* followStart = reinterpret_cast<const uint32_t*>(smem + hdr[NF_OFF_FOLLOWSTART]);
    const uint4* paths = reinterpret_cast<const uint4*>(smem + hdr[NF_OFF_PATHS]);

    const uint32_t wave = tid >> 6, lane = tid & 63;
    // per-wave scratch: best[nPos] then 4 x 64 words (newPos, newSrc, newTagsLo, newTagsHi)
    const uint32_t scratchWords = ((nPos + 3) & ~3u) + 256;
    uint32_t* best = reinterpret_cast<uint32_t*>(smem + blobBytes) + wave * scratchWords;
    uint32_t* newPos = best + ((nPos + 3) & ~3u);
    uint32_t* newSrc = newPos + 64;
    uint32_t* newTagsLo = newSrc + 64;
    uint32_t* newTagsHi = newTagsLo + 64;
    for (uint32_t i = lane; i < nPos; i += 64) best[i] = 0xFFFFFFFFu;
    waveLdsSync();

    const uint32_t line = blockIdx.x * kNfaWaves + wave;
    if (line >= nLines) return;  // wave-uniform
    const uint32_t o = off[line];
    const uint32_t L = len ? len[line] : off[line + 1] - o - sepBytes;

    int32_t cap[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) cap[s] = -1;
    uint32_t nThreads = 1;
    uint32_t myPos = nPos;  // lane 0: the start pseudo-position
    bool overflow = false;

    const uintptr_t addr = reinterpret_cast<uintptr_t>(data) + o;
    const uint32_t head = uint32_t(addr & 3);
    const uint32_t* words = reinterpret_cast<const uint32_t*>(addr - head);
    const uint32_t nWords = L ? (head + L + 3) / 4 : 0;
    uint32_t curWord = (lane < nWords) ? words[lane] : 0;
    int prevByte = -1;

    for (uint32_t i = 0; i < L && nThreads; ++i) {
        const uint32_t idx = head + i;
        if (i && (idx & 255u) == 0) {  // next 256-byte chunk: one coalesced dword per lane
            const uint32_t w = (idx >> 2) + lane;
            curWord = (w < nWords) ? words[w] : 0;
        }
        const uint32_t wsel = __builtin_amdgcn_readlane(curWord, (idx >> 2) & 63u);
        const int b = int((wsel >> ((idx & 3u) * 8)) & 0xFFu);
        const uint32_t cls = classMap[b];
        const uint32_t ctrue = condsTrue(prevByte, b);
        prevByte = b;

        const bool liveLane = lane < nThreads;
        const uint32_t fs = liveLane ? followStart[myPos] : 0;
        const uint32_t cnt = liveLane ? followStart[myPos + 1] - fs : 0;
        uint32_t totalCand;
        const uint32_t rankBase = waveExclusiveScan(cnt, lane, totalCand);
        // pass 1: filter candidates, elect per-target winners
        uint64_t passMask = 0;
        for (uint32_t k = 0; __any(k < cnt); ++k) {
            if (k < cnt) {
                const uint4 p = paths[fs + k];
                if (p.x != NF_TARGET_MATCH && (p.y & ~ctrue) == 0) {
                    const uint2 pm = posMask[p.x];
                    const uint32_t bit = cls < 32 ? (pm.x >> cls) & 1u : (pm.y >> (cls - 32)) & 1u;
                    if (bit) {
                        passMask |= uint64_t(1) << k;
                        atomicMin(&best[p.x], rankBase + k);
                    }
                }
            }
        }
        waveLdsSync();
        // pass 2: winners, in (lane, k) order
        uint64_t winMask = 0;
        for (uint64_t m = passMask; __any(m != 0);) {
            if (m) {
                const uint32_t k = uint32_t(__ffsll((long long)m)) - 1;
                m &= m - 1;
                const uint4 p = paths[fs + k];
                if (best[p.x] == rankBase + k) winMask |= uint64_t(1) << k;
            }
        }
        uint32_t totalWins;
        uint32_t slot = waveExclusiveScan(uint32_t(__popcll(winMask)), lane, totalWins);
        if (totalWins > 64) {
            overflow = true;
            break;
        }
        for (uint64_t m = winMask; __any(m != 0);) {
            if (m) {
                const uint32_t k = uint32_t(__ffsll((long long)m)) - 1;
                m &= m - 1;
                const uint4 p = paths[fs + k];
                newPos[slot] = p.x;
                newSrc[slot] = lane;
                newTagsLo[slot] = p.z;
                newTagsHi[slot] = p.w;
                best[p.x] = 0xFFFFFFFFu;
                ++slot;
            }
        }
        waveLdsSync();
        nThreads = totalWins;
        uint32_t src = lane;
        uint64_t tags = 0;
        if (lane < nThreads) {
            myPos = newPos[lane];
            src = newSrc[lane];
            tags = uint64_t(newTagsLo[lane]) | (uint64_t(newTagsHi[lane]) << 32);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (uint32_t(s) < nSlots) {
                const int32_t v = __shfl(cap[s], int(src), 64);
                cap[s] = ((tags >> s) & 1) ? int32_t(i) : v;
            }
        }
        waveLdsSync();  // newPos/newSrc are rewritten by the next byte's scatter
    }

    // acceptance at end of input: first thread (priority order) with a MATCH path whose assertions hold
    bool accept = false;
    uint64_t endTags = 0;
    if (!overflow && lane < nThreads) {
        const uint32_t ctrue = condsTrue(prevByte, -1);
        const uint32_t fs = followStart[myPos], fe = followStart[myPos + 1];
        for (uint32_t q = fs; q < fe; ++q) {
            const uint4 p = paths[q];
            if (p.x == NF_TARGET_MATCH && (p.y & ~ctrue) == 0) {
                accept = true;
                endTags = uint64_t(p.z) | (uint64_t(p.w) << 32);
                break;
            }
        }
    }
    const uint64_t acc = __ballot(accept);
    const bool matched = acc != 0;
    const uint32_t winner = matched ? uint32_t(__ffsll((long long)acc)) - 1 : 0;
    int32_t* out = caps + size_t(line) * 2 * nGroupsOut;
    if (lane == winner) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (uint32_t(s) < 2 * nGroupsOut) {
                int32_t v = -1;
                if (matched && uint32_t(s) < nSlots) v = ((endTags >> s) & 1) ? int32_t(L) : cap[s];
                out[s] = v;
            }
        }
        status[line] = overflow ? LC_OVERFLOW : (matched ? LC_MATCH : LC_NOMATCH);
    }
    for (uint32_t s = NS + lane; s < 2 * nGroupsOut; s += 64) out[s] = -1;
}

// ------------------------------------------------------------------------------------------------ device tables
static int ensureUploaded(lc_regex* re, int dev, bool tdfa, void** out) {
    std::lock_guard<std::mutex> g(re->deviceMutex);
    void** slot = tdfa ? &re->dTdfaBlob[dev] : &re->dNfaBlob[dev];
    if (!*slot) {
        const std::vector<uint32_t>& blob = tdfa ? re->tdfaBlob : re->nfaBlob;
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, blob.size() * 4));
        hipError_t e = hipMemcpy(p, blob.data(), blob.size() * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(p);
            return hipFail(e, "hipMemcpy(tables)");
        }
        *slot = p;
    }
    *out = *slot;
    return LC_OK;
}

void lcReleaseDeviceTables(lc_regex* re) {
    int cur = 0;
    bool haveCur = hipGetDevice(&cur) == hipSuccess;
    for (int d = 0; d < kLcMaxDevices; ++d) {
        if (re->dTdfaBlob[d] || re->dNfaBlob[d]) {
            if (hipSetDevice(d) == hipSuccess) {
                if (re->dTdfaBlob[d]) (void)hipFree(re->dTdfaBlob[d]);
                if (re->dNfaBlob[d]) (void)hipFree(re->dNfaBlob[d]);
            }
            re->dTdfaBlob[d] = re->dNfaBlob[d] = nullptr;
        }
    }
    if (haveCur) (void)hipSetDevice(cur);
}

template <int BLOCK>
static int launchTdfaBlock(const void* dBlob, uint32_t blobBytes, size_t lds, const uint8_t* d_data,
                           const uint32_t* d_off, const uint32_t* d_len, uint32_t sep, uint32_t n, uint32_t ngroups,
                           int32_t* d_caps, uint8_t* d_status, hipStream_t stream) {
    static thread_local size_t ldsAttrSet = 0;
    if (lds > 64 * 1024 && lds > ldsAttrSet) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&tdfa_match_kernel<BLOCK>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        ldsAttrSet = lds;
    }
    const uint32_t grid = (n + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(tdfa_match_kernel<BLOCK>, dim3(grid), dim3(BLOCK), lds, stream, d_data, d_off, d_len, sep, n,
                       static_cast<const uint32_t*>(dBlob), blobBytes, ngroups, d_caps, d_status);
    HIP_TRY(hipGetLastError());
    return LC_OK;
}

static int launchTdfa(lc_regex* re, int dev, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len,
                      uint32_t sep, uint32_t n, uint32_t ngroups, int32_t* d_caps, uint8_t* d_status,
                      hipStream_t stream) {
    void* dBlob = nullptr;
    int rc = ensureUploaded(re, dev, true, &dBlob);
    if (rc != LC_OK) return rc;
    const uint32_t blobBytes = uint32_t(re->tdfaBlob.size() * 4);
    const int block = lcTdfaPickBlock(blobBytes, re->tdfa.nRegs);
    if (block == 0) {
        tlsError = "tdfa tables + registers exceed LDS";
        return LC_ERR_UNSUPPORTED;
    }
    const size_t lds = lcTdfaLdsBytes(blobBytes, re->tdfa.nRegs, block);
    switch (block) {
        case 256: return launchTdfaBlock<256>(dBlob, blobBytes, lds, d_data, d_off, d_len, sep, n, ngroups, d_caps, d_status, stream);
        case 128: return launchTdfaBlock<128>(dBlob, blobBytes, lds, d_data, d_off, d_len, sep, n, ngroups, d_caps, d_status, stream);
        default: return launchTdfaBlock<64>(dBlob, blobBytes, lds, d_data, d_off, d_len, sep, n, ngroups, d_caps, d_status, stream);
    }
}

template <int NS>
static int launchNfaSlots(const void* dBlob, uint32_t blobBytes, size_t lds, const uint8_t* d_data,
                          const uint32_t* d_off, const uint32_t* d_len, uint32_t sep, uint32_t n, uint32_t ngroups,
                          int32_t* d_caps, uint8_t* d_status, hipStream_t stream) {
    static thread_local size_t ldsAttrSet = 0;
    if (lds > 64 * 1024 && lds > ldsAttrSet) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&nfa_match_kernel<NS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        ldsAttrSet = lds;
    }
    const uint32_t grid = (n + kNfaWaves - 1) / kNfaWaves;
    hipLaunchKernelGGL(nfa_match_kernel<NS>, dim3(grid), dim3(kNfaBlock), lds, stream, d_data, d_off, d_len, sep, n,
                       static_cast<const uint32_t*>(dBlob), blobBytes, ngroups, d_caps, d_status);
    HIP_TRY(hipGetLastError());
    return LC_OK;
}

static int launchNfa(lc_regex* re, int dev, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len,
                     uint32_t sep, uint32_t n, uint32_t ngroups, int32_t* d_caps, uint8_t* d_status,
                     hipStream_t stream) {
    if (re->nfaBlob.empty()) {
        tlsError = "pattern has no NFA program";
        return LC_ERR_UNSUPPORTED;
    }
    void* dBlob = nullptr;
    int rc = ensureUploaded(re, dev, false, &dBlob);
    if (rc != LC_OK) return rc;
    const uint32_t blobBytes = uint32_t(re->nfaBlob.size() * 4);
    const size_t lds = lcNfaLdsBytes(blobBytes, uint32_t(re->nfa.positions.size()));
    if (lds > 160 * 1024) {
        tlsError = "nfa tables exceed LDS";
        return LC_ERR_UNSUPPORTED;
    }
    const int slots = re->nfa.slotCount();
    if (slots <= 8) return launchNfaSlots<8>(dBlob, blobBytes, lds, d_data, d_off, d_len, sep, n, ngroups, d_caps, d_status, stream);
    if (slots <= 16) return launchNfaSlots<16>(dBlob, blobBytes, lds, d_data, d_off, d_len, sep, n, ngroups, d_caps, d_status, stream);
    if (slots <= 32) return launchNfaSlots<32>(dBlob, blobBytes, lds, d_data, d_off, d_len, sep, n, ngroups, d_caps, d_status, stream);
    return launchNfaSlots<64>(dBlob, blobBytes, lds, d_data, d_off, d_len, sep, n, ngroups, d_caps, d_status, stream);
}

static int matchOnStream(lc_regex* re, int engine, int dev, const uint8_t* d_data, const uint32_t* d_off,
                         const uint32_t* d_len, uint32_t sep, uint32_t n, uint32_t ngroups, int32_t* d_caps,
                         uint8_t* d_status, hipStream_t stream) {
    if (engine == LC_ENGINE_TDFA) {
        if (!re->hasTdfa) {
            tlsError = "pattern has no TDFA: " + re->tdfaError;
            return LC_ERR_UNSUPPORTED;
        }
        return launchTdfa(re, dev, d_data, d_off, d_len, sep, n, ngroups, d_caps, d_status, stream);
    }
    return launchNfa(re, dev, d_data, d_off, d_len, sep, n, ngroups, d_caps, d_status, stream);
}

extern "C" int lc_regex_match_device_engine(lc_regex_t* re, int engine, const uint8_t* d_data, const uint32_t* d_off,
                                            const uint32_t* d_len, uint32_t sep_bytes, uint32_t n, uint32_t ngroups,
                                            int32_t* d_caps, uint8_t* d_status, void* stream) {
    if (!re) return LC_ERR_ARG;
    if (n == 0) return LC_OK;
    if (!d_data || !d_off || !d_caps || !d_status) return LC_ERR_ARG;
    if (lc_device_count() <= 0) {
        tlsError = "no HIP device";
        return LC_ERR_NO_DEVICE;
    }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev >= kLcMaxDevices) return LC_ERR_ARG;
    if (engine == LC_ENGINE_AUTO) engine = re->engine;
    return matchOnStream(re, engine, dev, d_data, d_off, d_len, sep_bytes, n, ngroups, d_caps, d_status,
                         static_cast<hipStream_t>(stream));
}

extern "C" int lc_regex_match_device(lc_regex_t* re, const uint8_t* d_data, const uint32_t* d_off,
                                     const uint32_t* d_len, uint32_t sep_bytes, uint32_t n, uint32_t ngroups,
                                     int32_t* d_caps, uint8_t* d_status, void* stream) {
    return lc_regex_match_device_engine(re, LC_ENGINE_AUTO, d_data, d_off, d_len, sep_bytes, n, ngroups, d_caps,
                                        d_status, stream);
}

// ------------------------------------------------------------------------------------------------ host batches
namespace {

struct Slot {
    uint8_t* hData = nullptr;  // pinned
    uint32_t* hOff = nullptr;
    uint32_t* hLen = nullptr;
    int32_t* hCaps = nullptr;
    uint8_t* hStatus = nullptr;
    uint8_t* dData = nullptr;
    uint32_t* dOff = nullptr;
    uint32_t* dLen = nullptr;
    int32_t* dCaps = nullptr;
    uint8_t* dStatus = nullptr;
    size_t dataCap = 0, lineCap = 0, capsCap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    // what is in flight
    uint32_t first = 0, count = 0;
    bool busy = false;
};

struct HostPipeline {
    int device = -1;
    Slot slots[2];
    ~HostPipeline() { release(); }
    void release() {
        if (device < 0) return;
        (void)hipSetDevice(device);
        for (auto& s : slots) {
            if (s.stream) (void)hipStreamSynchronize(s.stream);
            (void)hipHostFree(s.hData); (void)hipHostFree(s.hOff); (void)hipHostFree(s.hLen);
            (void)hipHostFree(s.hCaps); (void)hipHostFree(s.hStatus);
            (void)hipFree(s.dData); (void)hipFree(s.dOff); (void)hipFree(s.dLen);
            (void)hipFree(s.dCaps); (void)hipFree(s.dStatus);
            if (s.done) (void)hipEventDestroy(s.done);
            if (s.stream) (void)hipStreamDestroy(s.stream);
            s = Slot();
        }
        device = -1;
    }
};

constexpr size_t kChunkBytes = 32u << 20;   // payload bytes per pipelined chunk
constexpr uint32_t kChunkLines = 1u << 18;  // and at most this many lines

int growSlot(Slot& s, size_t dataBytes, size_t lines, size_t capsInts) {
    if (!s.stream) {
        HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    if (dataBytes > s.dataCap) {
        (void)hipHostFree(s.hData); (void)hipFree(s.dData);
        s.hData = nullptr; s.dData = nullptr; s.dataCap = 0;
        size_t cap = dataBytes + (dataBytes >> 2) + 4096;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.hData), cap, hipHostMallocDefault));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.dData), cap));
        s.dataCap = cap;
    }
    if (lines > s.lineCap) {
        (void)hipHostFree(s.hOff); (void)hipHostFree(s.hLen); (void)hipHostFree(s.hStatus);
        (void)hipFree(s.dOff); (void)hipFree(s.dLen); (void)hipFree(s.dStatus);
        s.hOff = s.hLen = nullptr; s.hStatus = nullptr; s.dOff = s.dLen = nullptr; s.dStatus = nullptr; s.lineCap = 0;
        size_t cap = lines + (lines >> 2) + 64;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.hOff), cap * 4, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.hLen), cap * 4, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.hStatus), cap, hipHostMallocDefault));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.dOff), cap * 4));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.dLen), cap * 4));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.dStatus), cap));
        s.lineCap = cap;
    }
    if (capsInts > s.capsCap) {
        (void)hipHostFree(s.hCaps); (void)hipFree(s.dCaps);
        s.hCaps = nullptr; s.dCaps = nullptr; s.capsCap = 0;
        size_t cap = capsInts + (capsInts >> 2) + 64;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.hCaps), cap * 4, hipHostMallocDefault));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.dCaps), cap * 4));
        s.capsCap = cap;
    }
    return LC_OK;
}

int drainSlot(Slot& s, uint32_t ngroups, int32_t* caps, uint8_t* status) {
    if (!s.busy) return LC_OK;
    HIP_TRY(hipEventSynchronize(s.done));
    std::memcpy(caps + size_t(s.first) * 2 * ngroups, s.hCaps, size_t(s.count) * 2 * ngroups * 4);
    std::memcpy(status + s.first, s.hStatus, s.count);
    s.busy = false;
    return LC_OK;
}

}  // namespace

extern "C" int lc_regex_match_host(lc_regex_t* re, const uint8_t* data, const uint32_t* off, const uint32_t* len,
                                   uint32_t n, uint32_t ngroups, int32_t* caps, uint8_t* status) {
    if (!re || !off || !len || !caps || !status || (!data && n)) return LC_ERR_ARG;
    if (n == 0) return LC_OK;
    if (lc_device_count() <= 0) {
        tlsError = "no HIP device";
        return LC_ERR_NO_DEVICE;
    }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    static thread_local HostPipeline pipe;
    if (pipe.device != dev) {
        pipe.release();
        pipe.device = dev;
    }
    uint32_t next = 0;
    int which = 0;
    int rc = LC_OK;
    while (next < n) {
        // carve a chunk: up to kChunkLines lines / kChunkBytes payload bytes
        uint32_t cnt = 0;
        size_t bytes = 0;
        while (next + cnt < n && cnt < kChunkLines && (cnt == 0 || bytes + len[next + cnt] <= kChunkBytes)) {
            bytes += len[next + cnt];
            ++cnt;
        }
        Slot& s = pipe.slots[which];
        if ((rc = drainSlot(s, ngroups, caps, status)) != LC_OK) return rc;
        // contiguous fast path: the chunk's lines sit back to back (what ProcessorSplitLogStringNative leaves)
        const uint32_t lo = off[next];
        const uint64_t hi = uint64_t(off[next + cnt - 1]) + len[next + cnt - 1];
        bool contiguous = hi >= lo && (hi - lo) <= bytes + 2ull * cnt;
        if (contiguous)
            for (uint32_t i = 1; i < cnt && contiguous; ++i) contiguous = off[next + i] >= off[next + i - 1];
        const size_t stageBytes = contiguous ? size_t(hi - lo) : bytes;
        if ((rc = growSlot(s, stageBytes + 16, cnt, size_t(cnt) * 2 * ngroups)) != LC_OK) return rc;
        if (contiguous) {
            std::memcpy(s.hData, data + lo, stageBytes);
            for (uint32_t i = 0; i < cnt; ++i) {
                s.hOff[i] = off[next + i] - lo;
                s.hLen[i] = len[next + i];
            }
        } else {
            size_t at = 0;
            for (uint32_t i = 0; i < cnt; ++i) {
                std::memcpy(s.hData + at, data + off[next + i], len[next + i]);
                s.hOff[i] = uint32_t(at);
                s.hLen[i] = len[next + i];
                at += len[next + i];
            }
        }
        HIP_TRY(hipMemcpyAsync(s.dData, s.hData, stageBytes, hipMemcpyHostToDevice, s.stream));
        HIP_TRY(hipMemcpyAsync(s.dOff, s.hOff, size_t(cnt) * 4, hipMemcpyHostToDevice, s.stream));
        HIP_TRY(hipMemcpyAsync(s.dLen, s.hLen, size_t(cnt) * 4, hipMemcpyHostToDevice, s.stream));
        rc = lc_regex_match_device(re, s.dData, s.dOff, s.dLen, 0, cnt, ngroups, s.dCaps, s.dStatus, s.stream);
        if (rc != LC_OK) return rc;
        HIP_TRY(hipMemcpyAsync(s.hCaps, s.dCaps, size_t(cnt) * 2 * ngroups * 4, hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(hipMemcpyAsync(s.hStatus, s.dStatus, cnt, hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(hipEventRecord(s.done, s.stream));
        s.first = next;
        s.count = cnt;
        s.busy = true;
        next += cnt;
        which ^= 1;
    }
    for (auto& s : pipe.slots)
        if ((rc = drainSlot(s, ngroups, caps, status)) != LC_OK) return rc;
    return LC_OK;
}
