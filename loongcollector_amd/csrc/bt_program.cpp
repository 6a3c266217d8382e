// bt_program.cpp -- tree -> backtracking program (bt_vm.hpp).  Replaces, for patterns with back-references, the compile half of
// boost::regex on the reference path (core/plugin/processor/ProcessorParseRegexNative.cpp:64-67; IsRegexValid,
// core/common/ParamExtractor.cpp:199-209): the order in which the alternatives of every choice are tried is what makes the result
// boost's -- SPLIT's first operand is the way perl_matcher takes first (the earlier alternative; one more iteration of a greedy
// repeat; the way out of a lazy one).
#include "bt_program.hpp"

#include <map>

#include "bt_vm.hpp"

namespace lcregex {
namespace {

constexpr size_t kBtMaxInstructions = 65536;

struct Builder {
    std::vector<uint32_t> code;  // 4 words per instruction
    std::vector<ByteSet> sets;
    std::map<ByteSet, uint32_t> setIds;
    uint32_t nLoop = 0;

    uint32_t setId(const ByteSet& s) {
        auto it = setIds.find(s);
        if (it != setIds.end()) return it->second;
        const uint32_t id = uint32_t(sets.size());
        sets.push_back(s);
        setIds.emplace(s, id);
        return id;
    }
    uint32_t here() const { return uint32_t(code.size() / 4); }
    uint32_t emit(uint32_t op, uint32_t flags = 0, uint32_t x = 0, uint32_t y = 0, uint32_t z = 0) {
        if (code.size() / 4 >= kBtMaxInstructions) throw RegexError("backtracking program over 65536 instructions: unsupported");
        const uint32_t at = here();
        code.push_back(op | (flags << 8));
        code.push_back(x);
        code.push_back(y);
        code.push_back(z);
        return at;
    }
    uint32_t& x(uint32_t at) { return code[size_t(at) * 4 + 1]; }
    uint32_t& y(uint32_t at) { return code[size_t(at) * 4 + 2]; }

    static bool nullable(const Node& n) {
        switch (n.kind) {
            case Node::Empty:
            case Node::Assert:
            case Node::Look:
            case Node::BackRef: return true;  // (a group that matched the empty string refers to the empty string)
            case Node::Set: return false;
            case Node::Cat:
                for (const auto& k : n.kids)
                    if (!nullable(*k)) return false;
                return true;
            case Node::Alt:
                for (const auto& k : n.kids)
                    if (nullable(*k)) return true;
                return false;
            case Node::Repeat: return n.min == 0 || nullable(*n.kids[0]);
            case Node::Group:
            case Node::Atomic: return nullable(*n.kids[0]);
            case Node::Cond: return nullable(*n.kids[0]) || nullable(*n.kids[1]);
        }
        return true;
    }

    void gen(const Node& n) {
        switch (n.kind) {
            case Node::Empty: break;
            case Node::Set: emit(BT_SET, 0, setId(n.set)); break;
            case Node::Cat:
                for (const auto& k : n.kids) gen(*k);
                break;
            case Node::Alt: {
                // a | b | c:  SPLIT(a, next) a JMP out; SPLIT(b, next) b JMP out; c
                std::vector<uint32_t> jumps;
                for (size_t i = 0; i < n.kids.size(); ++i) {
                    if (i + 1 < n.kids.size()) {
                        const uint32_t sp = emit(BT_SPLIT);
                        x(sp) = here();
                        gen(*n.kids[i]);
                        jumps.push_back(emit(BT_JMP));
                        y(sp) = here();
                    } else {
                        gen(*n.kids[i]);
                    }
                }
                for (uint32_t j : jumps) x(j) = here();
                break;
            }
            case Node::Group:
                if (n.runCapture) throw RegexError("run captures on the backtracking engine: unsupported");
                if (n.capture) emit(BT_SAVE, 0, uint32_t(2 * n.capture));
                gen(*n.kids[0]);
                if (n.capture) emit(BT_SAVE, 0, uint32_t(2 * n.capture + 1));
                break;
            case Node::Atomic:
                emit(BT_ATOM_BEGIN);
                gen(*n.kids[0]);
                emit(BT_ATOM_END);
                break;
            case Node::Assert:
                if (!n.behindSeq.empty()) throw RegexError("undecided look-behind sequence in the tree: unsupported");  // (the parser decides or converts them)
                if (!n.aheadSeq.empty()) {  // a look-ahead over a sequence of byte classes (a window to the automata): the body in place
                    const uint32_t begin = emit(n.aheadNegative ? BT_NLOOK_BEGIN : BT_LOOK_BEGIN);
                    for (const ByteSet& b : n.aheadSeq) emit(BT_SET, 0, setId(b));
                    emit(n.aheadNegative ? BT_NLOOK_END : BT_LOOK_END);
                    x(begin) = here();
                    break;
                }
                emit(BT_ASSERT, (n.look.behind ? 1u : 0u) | (n.look.edgeOk ? 2u : 0u), setId(n.look.set));
                break;
            case Node::Look: {
                const uint32_t begin = emit(n.aheadNegative ? BT_NLOOK_BEGIN : BT_LOOK_BEGIN);
                if (n.look.behind && n.min > 0) emit(BT_BACK, 0, uint32_t(n.min));
                gen(*n.kids[0]);
                emit(n.aheadNegative ? BT_NLOOK_END : BT_LOOK_END);
                x(begin) = here();  // where a negative assertion that holds goes on
                break;
            }
            case Node::BackRef: emit(BT_BACKREF, 0, uint32_t(n.capture)); break;
            case Node::Cond: {
                const uint32_t c = emit(BT_COND, 0, uint32_t(n.capture));
                gen(*n.kids[0]);
                const uint32_t j = emit(BT_JMP);
                y(c) = here();
                gen(*n.kids[1]);
                x(j) = here();
                break;
            }
            case Node::Repeat: genRepeat(n); break;
        }
    }

    void genRepeat(const Node& n) {
        const Node& body = *n.kids[0];
        if (body.kind == Node::Set) {  // counted: one stack record per repeat
            emit(BT_REPSET, n.greedy ? 1u : 0u, setId(body.set), uint32_t(n.min), n.max < 0 ? BT_INF : uint32_t(n.max));
            return;
        }
        for (int i = 0; i < n.min; ++i) gen(body);
        if (n.max < 0) {
            const bool guard = nullable(body);
            const uint32_t reg = guard ? nLoop++ : 0;
            const uint32_t sp = emit(BT_SPLIT);
            const uint32_t start = here();
            if (guard) emit(BT_MARK, 0, reg);
            gen(body);
            uint32_t chk = BT_NONE;
            if (guard) chk = emit(BT_CHK, 0, reg);
            emit(BT_JMP, 0, sp);
            const uint32_t out = here();
            if (chk != BT_NONE) y(chk) = out;
            x(sp) = n.greedy ? start : out;
            y(sp) = n.greedy ? out : start;
        } else {
            std::vector<uint32_t> splits;
            for (int i = n.min; i < n.max; ++i) {
                const uint32_t sp = emit(BT_SPLIT);
                splits.push_back(sp);
                (n.greedy ? x(sp) : y(sp)) = here();
                gen(body);
            }
            for (uint32_t sp : splits) (n.greedy ? y(sp) : x(sp)) = here();
        }
    }
};

}  // namespace

std::vector<uint32_t> buildBtProgram(const ParsedRegex& re, bool icase) {
    Builder b;
    b.emit(BT_SAVE, 0, 0);
    b.gen(*re.root);
    b.emit(BT_SAVE, 0, 1);
    b.emit(BT_MATCH);
    std::vector<uint32_t> blob(BT_HEADER_WORDS, 0);
    blob[BT_NINST] = b.here();
    blob[BT_NSETS] = uint32_t(b.sets.size());
    blob[BT_NCAPS] = uint32_t(2 * (re.groupCount + 1));
    blob[BT_NLOOP] = b.nLoop;
    blob[BT_FLAGS] = icase ? 1u : 0u;
    blob[BT_OFF_SETS] = uint32_t(blob.size());
    for (const ByteSet& s : b.sets)
        for (int w = 0; w < 4; ++w) {
            blob.push_back(uint32_t(s.w[size_t(w)]));
            blob.push_back(uint32_t(s.w[size_t(w)] >> 32));
        }
    blob[BT_OFF_CODE] = uint32_t(blob.size());
    blob.insert(blob.end(), b.code.begin(), b.code.end());
    return blob;
}

}  // namespace lcregex
