// device_tables.h -- layout of the table blobs the host compilers hand to the gfx950 kernels.
// Shared by host C++ (packers) and HIP device code (readers).  All offsets are in bytes from the blob start and
// are multiples of 16 so the blob can be staged into LDS with 16-byte copies.
#pragma once
#include <stdint.h>

// ---------------------------------------------------------------- TDFA blob (engine LC_ENGINE_TDFA)
// header: 16 x u32
enum {
    TD_MAGIC = 0,        // 'TDFA'
    TD_NSTATES = 1,
    TD_NCLASSES = 2,
    TD_NREGS = 3,        // per-line offset registers
    TD_NSLOTS = 4,       // 2 * capture groups
    TD_START_ROW = 5,    // byte offset of the start state's row inside `trans`
    TD_OFF_CLASSMAP = 6, // u16[256]: class(b) * 4
    TD_OFF_TRANS = 7,    // u32[nStates*(nClasses+1)]: bits 0..19 next row byte offset, bits 20..31 register program (below)
    TD_OFF_FINALID = 8,  // u16[nStates]: 0xFFFF = not accepting
    TD_OFF_FINALMAP = 9, // u8[nFinal*nSlots]
    TD_OFF_OPSSTART = 10, // u32[nLists+1] index (in u16 units) into ops
    TD_OFF_OPS = 11,     // u16[]: n, then n x (dst | src<<8)
    TD_TOTAL_BYTES = 12,
    TD_ROW_BYTES = 13,   // (nClasses+1)*4: every row has one extra "identity" column (stay in this state, no register
                         // program) that lanes use for bytes outside their line, so the byte loop needs no branches
    TD_ID_COL = 14,      // byte offset of the identity column inside a row (= nClasses*4)
    TD_HEADER_WORDS = 16
};
#define TD_MAGIC_VALUE 0x41464454u
#define TD_ROW_MASK 0xFFFFFu
#define TD_LIST_SHIFT 20
// register program field h = entry >> 20:
//   0                      : nothing to do
//   0x800 | dst            : regs[dst] = pos                      (the overwhelmingly common case: one group boundary)
//   0x800 | 0x100 | dst    : regs[dst] = regs[dst+1] = pos        (a group closes and the next opens at the same offset)
//   1..0x7FF               : id of a general move list in `ops`
#define TD_OP_INLINE 0x800u
#define TD_OP_PAIR 0x100u
#define TD_MAX_LISTS 0x7FFu
#define TD_REG_POS 0xFFu
#define TD_REG_NONE 0xFEu

// ---------------------------------------------------------------- NFA blob (engine LC_ENGINE_NFA)
// header: 16 x u32
enum {
    NF_MAGIC = 0,         // 'NFA1'
    NF_NPOS = 1,          // positions (byte-consuming steps); start pseudo-position has index nPos
    NF_NSLOTS = 2,
    NF_NCLASSES = 3,
    NF_OFF_CLASSMAP = 4,  // u8[256] byte -> class
    NF_OFF_POSMASK = 5,   // u64[nPos] (as 2 x u32): bit c set iff position accepts byte class c  (nClasses <= 64)
    NF_OFF_FOLLOWSTART = 6, // u32[nPos+2]: follow list of position p = paths[followStart[p] .. followStart[p+1])
    NF_OFF_PATHS = 7,     // 4 x u32 per path: target (0xFFFFFFFF = MATCH), cond bits, tags lo, tags hi
    NF_TOTAL_BYTES = 8,
    NF_NPATHS = 9,
    NF_CONDS_USED = 10,
    NF_HEADER_WORDS = 16
};
#define NF_MAGIC_VALUE 0x3141464Eu
#define NF_TARGET_MATCH 0xFFFFFFFFu
