// follow_nfa.hpp -- epsilon-free, priority-ordered, capture-tagged NFA ("follow NFA").
//
// This is the merged-NFA form the north-star names: every *position* is one byte-consuming step of the regex;
// follow[p] lists, in leftmost-first (Perl/boost backtracking) priority order, every position reachable after
// consuming p, together with the capture slots written on the way (tag set) and the zero-width assertions that
// must hold (cond set).  Both device engines are driven by it:
//   * the NFA kernel keeps one wavefront lane per live thread and walks these lists directly;
//   * the TDFA builder determinises it (tdfa.hpp).
// Semantics restated: boost::regex_match leftmost-first backtracking == "highest-priority thread that is in
// MATCH when the input is exhausted" (core/common/StringTools.cpp:183-211 is the reference call site).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "regex_ast.hpp"

namespace lcregex {

constexpr int kMaxGpuGroups = 32;  // capture slots are carried in a 64-bit tag mask
constexpr int kMatchTarget = -1;

// which facts about the neighbouring bytes an assertion set needs
enum PrevCtxBits : uint8_t { kPrevAtStart = 1, kPrevWord = 2, kPrevSep = 4, kPrevCR = 8 };

struct ByteProps {  // facts about one byte (or about END / START)
    bool boundary = false;  // true for START (as prev) or END (as next)
    bool word = false, sep = false, cr = false, lf = false;
    static ByteProps of(unsigned c) {
        ByteProps p;
        p.word = isWordByte(c);
        p.sep = isLineSeparator(c);
        p.cr = c == '\r';
        p.lf = c == '\n';
        return p;
    }
    static ByteProps edge() {
        ByteProps p;
        p.boundary = true;
        return p;
    }
};

// true iff every assertion in `cond` (bit i = AssertKind i) holds between prev and next
bool condHolds(uint16_t cond, const ByteProps& prev, const ByteProps& next);
// PrevCtxBits an assertion set can observe
uint8_t condPrevNeeds(uint16_t cond);

struct FollowPath {
    int target;       // position index, or kMatchTarget
    uint64_t tags;    // bit s: capture slot s is written at the current offset
    uint16_t cond;    // bit k: AssertKind k must hold at the current offset
};

struct FollowNfa {
    int groupCount = 0;
    std::vector<std::string> groupNames;
    std::vector<ByteSet> positions;               // byte set consumed by each position
    std::vector<std::vector<FollowPath>> follow;  // follow[p]; follow[positions.size()] = start paths
    uint16_t condsUsed = 0;
    int slotCount() const { return 2 * groupCount; }
    int startIndex() const { return int(positions.size()); }
};

// Throws RegexError for constructs the device engines cannot honour bit-exactly (nullable loop bodies,
// more than kMaxGpuGroups groups, path explosion).
FollowNfa buildFollowNfa(const ParsedRegex& re);

}  // namespace lcregex
