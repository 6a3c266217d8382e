// follow_nfa.hpp -- epsilon-free, priority-ordered, capture-tagged NFA ("follow NFA").
//
// This is the merged-NFA form the north-star names: every *position* is one byte-consuming step of the regex;
// follow[p] lists, in leftmost-first (Perl/boost backtracking) priority order, every position reachable after
// consuming p, together with the capture slots written on the way (tag set) and the zero-width assertions that
// must hold (cond set: bit i = asserts[i], each a one-byte look behind or ahead).  Both device engines are driven
// by it:
//   * the NFA kernel keeps one wavefront lane per live thread and walks these lists directly;
//   * the TDFA builder determinises it (tdfa.hpp).
// Semantics restated: boost::regex_match leftmost-first backtracking == "highest-priority thread that is in
// MATCH when the input is exhausted" (core/common/StringTools.cpp:183-211 is the reference call site).
#pragma once

#include <cstdint>
#include <array>
#include <string>
#include <vector>

#include "regex_ast.hpp"

namespace lcregex {

constexpr int kMaxGpuGroups = 160;  // 320 capture slots (the widest NFA kernel instance carries 320 offsets per thread)
struct TagSet {                     // bit s: capture slot s
    std::array<uint64_t, 5> w{{0, 0, 0, 0, 0}};
    void set(int s) { w[size_t(s) >> 6] |= uint64_t(1) << (s & 63); }
    bool test(int s) const { return (w[size_t(s) >> 6] >> (s & 63)) & 1; }
    bool any() const { return (w[0] | w[1] | w[2] | w[3] | w[4]) != 0; }
    uint32_t word32(size_t k) const { return k < 10 ? uint32_t(w[k >> 1] >> (32 * (k & 1))) : 0u; }
    bool operator==(const TagSet& o) const { return w == o.w; }
    bool operator!=(const TagSet& o) const { return !(w == o.w); }
    bool operator<(const TagSet& o) const { return w < o.w; }
};
constexpr int kMaxAsserts = 32;     // distinct one-byte look assertions per pattern (cond mask is 32 bits)
constexpr int kMatchTarget = -1;
constexpr int kAssertEvent = 20000;
constexpr int kEdge = -1;           // "byte" value standing for START (behind) / END (ahead)

struct FollowPath {
    int target;       // position index, or kMatchTarget
    TagSet tags;      // bit s: capture slot s is written at the current offset
    uint32_t cond;    // bit i: asserts[i] must hold at the current offset
    // Only for patterns with atomic groups: what the epsilon path crosses, in order.  code +(g+1) = enter atomic
    // group instance g, -(g+1) = leave it, kAssertEvent+i = assertion i is tested here (so that "which exits happened
    // before an assertion failed" is known).  `visit` numbers the traversal of an exit: two paths of one follow list
    // that share the same exit visit left the group through the very same body match and differ only afterwards.
    struct Event {
        int32_t code;
        int32_t visit;
        bool operator==(const Event& o) const { return code == o.code && visit == o.visit; }
    };
    std::vector<Event> atoms;
};

struct FollowNfa {
    int groupCount = 0;
    std::vector<std::string> groupNames;
    std::vector<ByteSet> positions;               // byte set consumed by each position
    std::vector<std::vector<FollowPath>> follow;  // follow[p]; follow[positions.size()] = start paths
    std::vector<LookAssert> asserts;              // the distinct look primitives cond bits refer to
    uint32_t condsUsed = 0;                       // union of all path conds
    uint32_t behindMask = 0;                      // bits of asserts that look behind
    int atomicCount = 0;                          // atomic group instances; > 0 means only the TDFA engine can run it
    // LC_SYNTAX_SEARCH wrapper (?s:.*?)(re)(?s:.*): the positions of its two '.' (else -1).  A search can RESUME in the
    // middle of a line (the next match of an iterate-all-matches caller): that is the state "the prefix position has
    // just consumed the byte before the resume point".
    int searchPrefix = -1, searchSuffix = -1;
    // groups written "(?=(S*))" (regex_ast.hpp Node::runCapture): 0-based group index and S.  The automata leave their
    // end slot unset; whoever reports captures sets end = begin + length of the run of S bytes at begin.
    std::vector<std::pair<int, ByteSet>> runGroups;
    int slotCount() const { return 2 * groupCount; }
    int startIndex() const { return int(positions.size()); }

    // which behind-assertions hold when the previous byte is `prev` (kEdge = start of input)
    uint32_t behindBits(int prev) const;
    // which ahead-assertions hold when the next byte is `next` (kEdge = end of input)
    uint32_t aheadBits(int next) const;
    bool condHolds(uint32_t cond, int prev, int next) const { return (cond & ~(behindBits(prev) | aheadBits(next))) == 0; }
};

// Throws RegexError for constructs the device engines cannot honour bit-exactly (nullable loop bodies,
// more than kMaxGpuGroups groups, more than kMaxAsserts look assertions, path explosion).
FollowNfa buildFollowNfa(const ParsedRegex& re);

}  // namespace lcregex
