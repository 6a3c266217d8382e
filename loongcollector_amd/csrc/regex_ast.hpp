// regex_ast.hpp -- Perl-syntax regex front-end for the MI355X parse processor (host side, C++17).
//
// Replaces the *compile* half of boost::regex on the reference hot path:
//   mReg.emplace_back(mRegex)            core/plugin/processor/ProcessorParseRegexNative.cpp:64-67
//   IsRegexValid(regex)                   core/common/ParamExtractor.cpp:199-209
// The accepted language is the subset of Boost.Regex Perl syntax that is regular, plus back-references (round 6: those
// patterns run on the device backtracking engine, bt_vm.hpp); recursion, conditionals and look-arounds of variable width are
// reported as RegexError so that plugin Init fails loudly, exactly where the reference would reject an invalid regex.
#pragma once

#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

namespace lcregex {

struct RegexError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// 256-bit byte set
struct ByteSet {
    std::array<uint64_t, 4> w{{0, 0, 0, 0}};
    void add(unsigned c) { w[(c >> 6) & 3] |= uint64_t(1) << (c & 63); }
    void addRange(unsigned lo, unsigned hi) {
        for (unsigned c = lo; c <= hi; ++c) add(c);
    }
    bool has(unsigned c) const { return (w[(c >> 6) & 3] >> (c & 63)) & 1; }
    void unite(const ByteSet& o) {
        for (int i = 0; i < 4; ++i) w[i] |= o.w[i];
    }
    void invert() {
        for (auto& x : w) x = ~x;
    }
    bool empty() const { return !(w[0] | w[1] | w[2] | w[3]); }
    bool operator==(const ByteSet& o) const { return w == o.w; }
    bool operator<(const ByteSet& o) const { return w < o.w; }
    static ByteSet all() {
        ByteSet s;
        s.invert();
        return s;
    }
};

// Syntax options; defaults reproduce `boost::regex(str)`: Perl syntax, '.' matches '\n' (mod_s), '^'/'$' match at
// embedded line separators (mod_m).
struct Syntax {
    bool icase = false;
    bool dotAll = true;
    bool multiLine = true;
    bool extended = false;
    bool namedOnly = false;  // unnamed (...) groups do not capture (Grok semantics: only named groups are emitted)
    // github.com/dlclark/regexp2 with the RE2 option (what the Go Grok plugin compiles with, processor_grok.go:343):
    // \s is RE2's [\t\n\f\r ] (no \v), and escapes that mean nothing to it -- \< \> \` \' -- are the literal character,
    // where boost's Perl syntax reads them as word-start/-end and buffer-start/-end assertions.
    bool regexp2 = false;
};

// A primitive zero-width assertion looks at ONE neighbouring byte:
//   behind: holds iff  (at start of input ? edgeOk : previous byte in set)
//   ahead : holds iff  (at end of input   ? edgeOk : next byte in set)
// Everything else is composed from these by the parser:  ^ $ \A \z \b \B \< \> and the one-byte look-arounds
// (?=[..]) (?![..]) (?<=[..]) (?<![..])  (a negative look-around is the complemented set with edgeOk = true).
struct LookAssert {
    bool behind = false;
    bool edgeOk = false;
    ByteSet set;
    bool operator==(const LookAssert& o) const { return behind == o.behind && edgeOk == o.edgeOk && set == o.set; }
};

struct Node {
    enum Kind : uint8_t { Empty, Set, Cat, Alt, Repeat, Group, Assert, Atomic, BackRef, Look, Cond } kind = Empty;
    std::vector<std::unique_ptr<Node>> kids;  // Cat/Alt: n children; Repeat/Group/Atomic: 1
    ByteSet set;                               // Set
    int min = 0, max = -1;                     // Repeat (max < 0: unbounded)
    bool greedy = true;                        // Repeat
    int capture = 0;                           // Group: 1-based capture index, 0 = non-capturing; BackRef: the group referred to
    LookAssert look;                           // Assert
    // parser-internal: a look-AHEAD whose body is a sequence of >= 2 character classes.  It never reaches the automata:
    // the parser drops it when what follows already implies it, and refuses the pattern otherwise (regex_parse.cpp).
    std::vector<ByteSet> aheadSeq;
    bool aheadNegative = false;
    // round 5: ... and when what follows does NOT decide it, the look-ahead stays in the tree as a WINDOW (window = true): the next
    // aheadSeq.size() bytes must (positive) / must not (negative) be in aheadSeq[0], aheadSeq[1], ...  buildFollowNfa turns windows into
    // a product of the follow NFA with the window's chain (follow_nfa.cpp): "(?= } ntoreturn:)" of the library's MONGO_QUERY.
    bool window = false;
    // parser-internal: a look-BEHIND whose body is a sequence of >= 2 character classes ("(?<={ )").  Decided by the parser from the
    // fixed-width sub-expressions in front of it, or the pattern is refused.
    std::vector<ByteSet> behindSeq;
    bool behindNegative = false;
    // Group written "(?=(S*))" -- a look-ahead that always holds and whose only effect is the capture: the group begins
    // at the current offset and runs over the bytes of `set` that follow (Grok's "(?=%{GREEDYDATA:message})").  The
    // automata stamp the BEGIN slot only; the end is a function of the begin and is filled in after the match
    // (gpu_runtime.hip run_capture_kernel).  kids[0] is an Empty node.
    bool runCapture = false;
    // Cond (round 6): "(?(N)yes|no)" -- capture = the group asked about, kids[0] = yes, kids[1] = no (Empty when absent); runs on the
    // device backtracking engine like Look (below)
    // Look (round 6): a GENERAL look-around -- kids[0] is the body, aheadNegative its sign, look.behind its direction, min the body's
    // fixed length for a look-behind (the walk steps that many bytes back and runs the body forward, as the backtracking engines do).
    // What the one-byte and class-sequence forms above do not cover ("(?=a+b)", "(?<=ab|cd)", a look-behind the text in front of it does
    // not decide): no automaton is built for such a tree, it runs on the device backtracking engine (bt_vm.hpp).
};

struct ParsedRegex {
    std::unique_ptr<Node> root;
    int groupCount = 0;                    // == boost::basic_regex::mark_count()
    std::vector<std::string> groupNames;   // [0] unused; "" when unnamed
    // round 6: the pattern holds back-references (\1 .. \N): not a regular language -- no automaton is built, the handle runs the
    // device backtracking engine (bt_program.hpp, bt_vm.hpp: LC_ENGINE_BT)
    bool hasBackRef = false;
    bool hasGeneralLook = false;           // ... or general look-arounds (Node::Look): the same engine
};

ParsedRegex parseRegex(std::string_view pattern, Syntax syntax = Syntax());

bool isWordByte(unsigned c);
bool isLineSeparator(unsigned c);  // \n \r \f  (Boost is_separator<char>)

}  // namespace lcregex
