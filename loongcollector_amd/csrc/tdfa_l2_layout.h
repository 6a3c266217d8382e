// tdfa_l2_layout.h -- blob of a tagged DFA whose tables stay in global memory (tdfa_l2_kernel.hpp); shared with the host packer.
#pragma once
enum {
    TL_MAGIC = 0,          // 'TDL2'
    TL_NSTATES = 1,
    TL_NCLASSES = 2,
    TL_NREGS = 3,          // offset registers per line
    TL_NSLOTS = 4,
    TL_START = 5,
    TL_OFF_TRANS = 6,      // byte offsets from the start of the blob: u32[nStates][nClasses], low16 = next state (0 = dead),
                           //   high16 = register program (0 = none)
    TL_OFF_OPSSTART = 7,   // u32[programs + 1]
    TL_OFF_OPS = 8,        // u16[]: per program n, then n words dst | src << 8 (src 0xFF = the current offset)
    TL_OFF_FINALID = 9,    // u16[nStates], 0xFFFF = not accepting
    TL_OFF_FINALMAP = 10,  // u8[finals][nSlots]: register | 0xFF = end of line | 0xFE = unset
    TL_OFF_STARTAFTER = 11, // u32[nClasses] or 0: state a resumed search starts in, by the class of the byte before
    TL_TOTAL_BYTES = 12,
    TL_ABSORB = 13,        // a state that accepts, moves to itself on every byte class and runs no register program (the search wrapper's
                           // suffix (?s:.*) on its own), or 0: a line that reaches it is decided -- the kernel stops reading it
    TL_OFF_QUIET = 14,     // u64[nStates]: bit c = byte class c (< 64) keeps the state and runs no register program -- the bytes the
                           // wave-per-value kernel crosses without a table read (tdfa_wave_kernel)
    TL_MISS = 15,          // lazy / partial automata (tdfa.hpp buildTdfaLazy): the state that stands for "this transition was never
                           // computed", or 0 (a complete automaton): a line that reaches it is NOT decided -- the kernel leaves it
                           // LC_PENDING and raises the launch's pending flag; the thread-list kernels behind take it from its first byte
    TL_HEADER_WORDS = 16   // the class map (u8[256]) follows the header
};
#define TL_MAGIC_VALUE 0x324C4454u

// ---- launch shapes shared with the host (gpu_runtime.hip, grok_device.hip)
#include <stdint.h>
constexpr int kTdfaWaveBlock = 256;                    // tdfa_wave_kernel: one value per wavefront
constexpr int kTdfaWaveValues = kTdfaWaveBlock / 64;   // ... four values per workgroup
// one job of tdfa_wave_multi_kernel (several automata, each over its own values, in one launch: tdfa_l2_kernel.hpp)
struct TdfaWaveJob {
    const uint32_t* blob;
    const uint32_t* off;
    const uint32_t* len;
    const uint32_t* resume;   // or nullptr
    int32_t* caps;
    uint8_t* status;
    uint32_t* missFlag;       // or nullptr
    uint32_t n, nGroupsOut, stageBytes, seq, missStatus, firstBlock;
};
constexpr uint32_t kTdfaWaveMaxJobs = 64;
