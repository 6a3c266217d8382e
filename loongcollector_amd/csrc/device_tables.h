// device_tables.h -- layout of the table blobs the host compilers hand to the gfx950 kernels.
// Shared by host C++ (packers) and HIP device code (readers).  All offsets are in bytes from the blob start and
// are multiples of 16 so the blob can be staged into LDS with 16-byte copies.  The blob is staged at LDS
// address 0 (the kernels use dynamic LDS only), so blob offsets are LDS addresses.
#pragma once
#include <stdint.h>

// ---------------------------------------------------------------- TDFA blob (engine LC_ENGINE_TDFA)
// fixed prefix:  [0,64) header   [64,320) class map   [320, ...) transition table
enum {
    TD_MAGIC = 0,        // 'TDFA'
    TD_NSTATES = 1,
    TD_NCLASSES = 2,
    TD_NREGS = 3,        // bits 0..15: per-line offset registers, INCLUDING the trailing dummy register;
                         // bits 16..28: offset / 16 of the FOLD WORDS (0 = none) -- u32 count, then per word
                         //   set register | member << 8 | member << 16 | member << 24 (0xFF = no member): at the end of a
                         //   line every member reads as max(member, set register): folded multi-stamp programs, regex_handle.cpp;
                         // bit 31 (TD_NREGS_NO_GENERAL): no transition carries a general register program.
                         // With fold words the offset registers are cleared to 0 at the start of a line.
    TD_NSLOTS = 4,       // 2 * capture groups
    TD_START_ROW = 5,    // LDS address of the start state's row (= TD_TRANS_OFFSET + start*rowBytes)
    TD_OFF_STARTAFTER = 6, // u32[nClasses]: row address to resume a search in, by class of the byte before the resume
                         // point (0 = the pattern is not a search pattern); the class map itself sits at TD_CMAP_OFFSET
    TD_OFF_PAIR = 7,     // 0, or offset of the byte-PAIR extension (4 x u32, TP_*): a second transition table indexed by
                         // (state, class of byte 2k, class of byte 2k+1) that the kernel walks two bytes per lookup
    TD_OFF_FINALID = 8,  // u16[nStates]: 0xFFFF = not accepting
    TD_OFF_FINALMAP = 9, // u8[nFinal*nSlots]
    TD_OFF_OPSSTART = 10, // u32[nLists+1] index (in u16 units) into ops
    TD_OFF_OPS = 11,     // u16[]: n, then n x (dst | src<<8)
    TD_TOTAL_BYTES = 12,
    TD_ROW_BYTES = 13,   // (nClasses+1)*4: every row has one extra "identity" column (stay in this state, stamp the
                         // dummy register) that lanes use for bytes outside their line
    TD_ID_COL = 14,      // byte offset of the identity column inside a row (= nClasses*4)
    TD_BLOCK = 15,       // workgroup size the register offsets were encoded for
    TD_HEADER_WORDS = 16
};
// byte-pair extension header (at TD_OFF_PAIR)
enum {
    TP_BASE = 0,       // LDS address of pair row 0 (the dead state): u32[nStates][(nClasses+1)^2]
    TP_ROW_BYTES = 1,  // (nClasses+1)^2 * 4
    TP_OFF_CMAPA = 2,  // u16[256]: class(b) * (nClasses+1) * 4 -- column offset contributed by the FIRST byte of a pair
                       // (the second byte contributes cmap8[b] = class * 4, the table at TD_CMAP_OFFSET)
    TP_ID_A = 3,       // nClasses * (nClasses+1) * 4: first-byte offset of the identity class
    TP_FORMAT = 4,     // 0: two stamps per pair entry (below); 1: ONE stamp per pair entry (TP1_*, LC_TDFA_PAIR=2)
    TP_OFF_DERIVE = 5, // format 1: offset of the DERIVE WORDS (0 = none): u32 count, then per word  b | a << 8 | delta << 16:
                       //   at the end of a line, AFTER the fold words, register b reads as register a + delta (regex_handle.cpp
                       //   planTdfaDerive: b is only ever stamped one byte behind a, so its own stamps were dropped)
    TP_HEADER_WORDS = 8
};
// pair entry, format 1 (one stamp per byte pair; tdfa_stream_kernel.hpp tdfaStreamPair1Chunk):
//   bits 0..15   LDS address of the next state's PAIR row
//   bits 16..22  rA: the register the pair stamps (index; the dummy register = "none")      bit 23  its value is pos + 1 (else pos)
//   bits 24..30  rB: DOUBLE entries only -- the second byte's register (value pos + 1; rA is then the first byte's, value pos)
//   bit 31       DOUBLE: both bytes stamp, different registers.  The kernel writes rA in line, and -- per chunk, only when some
//                lane of the wavefront met a DOUBLE -- rB as max(register, pos + 1) behind the chunk's rA stamps
#define TP1_DELTA 0x00800000u
#define TP1_DOUBLE 0x80000000u
// pair entry: bits 0..15 LDS address of the next state's PAIR row; bits 16..23 / 24..31 register stamped by the first /
// second byte (index; the dummy register = "none"); bit 7 of either index set = that byte carries a general register
// program -> the chunk is replayed byte by byte on the single-byte table
#define TP_GENERAL 0x80u
#define TP_MAX_TABLE_BYTES (26u << 10)
#define TD_MAGIC_VALUE 0x41464454u
#define TD_CMAP_OFFSET 64u    // u8[256]: class(b) * 4  (column byte offset inside a row; at most 63 classes)
#define TD_TRANS_OFFSET 320u  // u32[nStates][nClasses+1]
// transition entry:
//   bits 0..15  LDS address of the next state's row (state 0 = dead, row address TD_TRANS_OFFSET)
//   bits 16..31 register program:
//       reg * BLOCK * 4          LDS offset (from regs[0][lane]) of the offset register that receives `pos`;
//                                transitions that stamp nothing name the dummy register (index nRegs-1)
//       (list << 1) | 1          general move list `list` in `ops` (anything that is not "one register = pos")
#define TD_OP_GENERAL 0x1u
#define TD_NREGS_NO_GENERAL 0x80000000u
#define TD_MAX_LISTS 0x7FFFu
#define TD_MAX_TABLE_END 0x10000u     // rows must be addressable with 16 bits
#define TD_MAX_REG_AREA 0x10000u      // nRegs*BLOCK*4: register offsets must fit the 16-bit field
#define TD_REG_POS 0xFFu
#define TD_REG_NONE 0xFEu

// ---------------------------------------------------------------- NFA blob (engine LC_ENGINE_NFA)
// header: NF_HEADER_WORDS x u32
enum {
    NF_MAGIC = 0,         // 'NFA1'
    NF_NPOS = 1,          // positions (byte-consuming steps); start pseudo-position has index nPos
    NF_NSLOTS = 2,
    NF_NCLASSES = 3,
    NF_OFF_CLASSMAP = 4,  // u8[256] byte -> class
    NF_OFF_POSMASK = 5,   // u32[nPos+1][maskWords]: bit c set iff position accepts byte class c
    NF_OFF_FOLLOWSTART = 6, // u32[nPos+2]: follow list of position p = paths[followStart[p] .. followStart[p+1])
    NF_OFF_PATHS = 7,     // 2 x u32 per path: x = target (16 bits, 0xFFFF = MATCH) | aux index << 16 (0 = no cond, no tags);
                          // y = (first event << 8) | event count (atomic patterns, else 0)
    NF_TOTAL_BYTES = 8,
    NF_NPATHS = 9,
    NF_CONDS_USED = 10,
    NF_OFF_STABLE = 11,   // u32[nPos+1][maskWords]: bit c set iff on byte class c the position's ONLY possible move is its own
                          // unconditional, tag-free self loop (the kernel's steady-state fast path); search patterns: moves
                          // of lower priority to the wrapper's suffix position do not count (regex_handle.cpp packNfaBlob)
    NF_OFF_BEHIND = 12,   // u32[nClasses+1]: look-behind assertions (cond bits) that hold when the previous byte has class
                          // c; entry nClasses = start of input
    NF_OFF_AHEAD = 13,    // u32[nClasses+1]: look-ahead assertions that hold when the next byte has class c; entry
                          // nClasses = end of input
    // patterns with atomic groups / possessive quantifiers only (all three 0 otherwise):
    NF_ATOMIC = 14,       // number of atomic group instances
    NF_OFF_EVENTS = 15,   // u32[]: low16 = code (int16: +(g+1) enter group instance g, -(g+1) leave it, 20000+i assertion i
                          // is tested here), high16 = exit visit (follow_nfa.hpp FollowPath::Event)
    NF_OFF_ATOMICPOS = 16, // u32[(nPos+1)/32+1]: bit p = some path out of position p enters or leaves an atomic group
    NF_OFF_AUX = 17,      // auxWords x u32 per entry: cond bits, then the tag words -- the distinct (cond, tags) of the paths
    NF_SEARCH = 18,       // 1: position 0 is the lazy prefix of the LC_SYNTAX_SEARCH wrapper (the kernel may skip ahead to the
                          // next byte the pattern can start with while only that thread is alive)
    NF_OFF_TOUCHY = 19,   // atomic patterns: u32[nPos+1][maskWords], bit c = on byte class c a thread on this position needs
                          // the ordered commit pass (a path leaves a group, or enters one towards a position that takes class c)
    NF_MASK_WORDS = 20,   // words per class mask: 2, or 4 for patterns with 65..128 byte classes
    NF_AUX_WORDS = 21,    // words per aux entry: 4 (cond + 2 tag words); 8 (cond + 4) for 65..128 capture slots; 16 (cond + 10) for
                          // 129..320
    NF_SUFFIX = 22,       // 1: the LAST position is the greedy suffix (?s:.*) of the search wrapper (LC_SYNTAX_SEARCH, with or without
                          // LC_SYNTAX_PREFIX): a thread on it takes every byte and ends on MATCH whatever follows, so nothing ranked
                          // below it can win, and a thread list that is that thread alone is decided -- the kernels stop there
    NF_OFF_QUASI = 23,    // 0, or the offset of the DOOMED-SPAWN tables (round 4): u32 rows, u32 offset of the rows, then u16 idx[nPos + 1]
                          // (0 = the position has no row, else row + 1); rows: u32[rows][nClasses][maskWords] -- bit d of row r, class c:
                          // a thread on the position, on a byte of class c FOLLOWED by a byte of class d, does nothing but repeat
                          // itself: its one clean self loop passes, and every other path that passes leads to a position none of
                          // whose follow paths takes d -- a thread that would live for exactly one byte and touch nothing
                          // (the " SA (SPI=" behind a GREEDYDATA tried at every space; the lazy search prefix tried at every byte a
                          // format can begin with).  Such a byte is as steady as one whose only move is the self loop.
    // Round 5: the follow lists BY BYTE CLASS.  A step's candidates are the (thread, follow path) pairs of the live threads, 64 per
    // election round -- and a thread inside an IP address alternation has a follow list of a hundred paths of which the byte takes two:
    // on CISCOFW313005 a step ran thirty rounds (13 000 cycles) to find a handful of survivors.  cstart[(p * nClasses + c)] ..
    // cstart[.. + 1] = the range in cpaths[] of the indices (into NF_OFF_PATHS, priority order kept) of position p's paths whose
    // target takes byte class c (MATCH paths take no byte: never listed).  Always read from global memory (L2), also by kernels that
    // stage the program in LDS: the tables lie BEHIND NF_STAGE_BYTES, the part of the blob those kernels stage.  0 = not packed
    // (the tables would exceed 16 MiB).
    NF_OFF_CSTART = 24,   // u32[(nPos + 1) * nClasses + 1]
    NF_OFF_CPATHS = 25,   // u32[]
    NF_STAGE_BYTES = 26,  // bytes of the blob in front of the class tables (a multiple of 16)
    NF_HEADER_WORDS = 28
};
#define NF_MAGIC_VALUE 0x3141464Eu
#define NF_TARGET_MATCH 0xFFFFFFFFu
