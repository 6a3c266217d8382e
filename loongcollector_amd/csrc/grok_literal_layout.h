// grok_literal_layout.h -- blob of the Grok literal index (u32 words): GL_* header, class map u8[256], output masks u64[nStates],
// table u16[nStates][nClasses] whose bit 15 says "the target state has an output" (so the mask table is touched only where a
// literal ends).  Shared by the host builder (grok_literal_index.cpp) and the kernel (grok_kernel.hpp).
#pragma once
enum { GL_NSTATES = 0, GL_NCLASSES = 1, GL_OFF_MASKS = 2, GL_OFF_TABLE = 3, GL_ALWAYS_LO = 4, GL_ALWAYS_HI = 5, GL_HEADER_WORDS = 8 };
