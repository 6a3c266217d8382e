// screen_kernel_layout.h -- blob layout of the global-memory screen DFA (screen_kernel.hpp); shared with the host packer.
#pragma once
enum {
    SC_MAGIC = 0,       // 'SCR1'
    SC_NSTATES = 1,
    SC_NCLASSES = 2,
    SC_START = 3,
    SC_SINK = 4,        // the state "a match is certain whatever follows" (0xFFFFFFFF: none)
    SC_OFF_ACCEPT = 5,  // byte offsets from the start of the blob
    SC_OFF_TABLE = 6,
    SC_TOTAL_BYTES = 7,
    SC_HEADER_WORDS = 8  // the class map follows the header
};
#define SC_MAGIC_VALUE 0x31524353u
