// bt_program.hpp -- compiles the parser's tree into the instruction program of the device backtracking engine (bt_vm.hpp).
#pragma once

#include <cstdint>
#include <vector>

#include "regex_ast.hpp"

namespace lcregex {
// Throws RegexError ("... unsupported ...") for trees the engine does not run: multi-byte look-around windows, run captures
// (Grok's "(?=(S*))" form), programs over 65 536 instructions.  icase: back-references compare ASCII-folded bytes (the byte classes
// were folded by the parser already).
std::vector<uint32_t> buildBtProgram(const ParsedRegex& re, bool icase);
}  // namespace lcregex
