// follow_nfa.cpp -- AST -> prioritised Thompson program -> epsilon-free follow NFA.
#include "follow_nfa.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace lcregex {

uint32_t FollowNfa::behindBits(int prev) const {
    uint32_t m = 0;
    for (size_t i = 0; i < asserts.size(); ++i)
        if (asserts[i].behind && (prev == kEdge ? asserts[i].edgeOk : asserts[i].set.has(unsigned(prev)))) m |= 1u << i;
    return m;
}
uint32_t FollowNfa::aheadBits(int next) const {
    uint32_t m = 0;
    for (size_t i = 0; i < asserts.size(); ++i)
        if (!asserts[i].behind && (next == kEdge ? asserts[i].edgeOk : asserts[i].set.has(unsigned(next)))) m |= 1u << i;
    return m;
}

namespace {

struct Inst {
    enum Op : uint8_t { Char, Split, Jump, Save, Assert, AtomEnter, AtomExit, Match } op;
    int x = 0, y = 0;  // Char: position index | Split: preferred, other | Jump: target | Save: slot | Assert: kind
};

bool nullable(const Node& n) {
    switch (n.kind) {
        case Node::Empty:
        case Node::Assert: return true;
        case Node::Set: return false;
        case Node::Cat:
            for (auto& k : n.kids)
                if (!nullable(*k)) return false;
            return true;
        case Node::Alt:
            for (auto& k : n.kids)
                if (nullable(*k)) return true;
            return false;
        case Node::Repeat: return n.min == 0 || nullable(*n.kids[0]);
        case Node::Group:
        case Node::Atomic: return nullable(*n.kids[0]);
    }
    return true;
}

class Builder {
public:
    std::vector<Inst> code;
    std::vector<ByteSet> positions;
    std::vector<LookAssert> asserts;
    int atomicCount = 0;
    std::vector<std::pair<int, ByteSet>> runGroups;

    int assertIndex(const LookAssert& a) {
        for (size_t i = 0; i < asserts.size(); ++i)
            if (asserts[i] == a) return int(i);
        if (int(asserts.size()) >= kMaxAsserts) throw RegexError("unsupported: too many distinct look assertions");
        asserts.push_back(a);
        return int(asserts.size()) - 1;
    }

    int emit(Inst::Op op, int x = 0, int y = 0) {
        if (code.size() > 200000) throw RegexError("regex too large after repeat expansion");
        code.push_back({op, x, y});
        return int(code.size()) - 1;
    }

    void gen(const Node& n) {
        switch (n.kind) {
            case Node::Empty: break;
            case Node::Set:
                positions.push_back(n.set);
                emit(Inst::Char, int(positions.size()) - 1);
                break;
            case Node::Cat:
                for (auto& k : n.kids) gen(*k);
                break;
            case Node::Alt: {
                std::vector<int> exits;
                for (size_t i = 0; i < n.kids.size(); ++i) {
                    if (i + 1 < n.kids.size()) {
                        int sp = emit(Inst::Split);
                        code[sp].x = int(code.size());
                        gen(*n.kids[i]);
                        exits.push_back(emit(Inst::Jump));
                        code[sp].y = int(code.size());
                    } else {
                        gen(*n.kids[i]);
                    }
                }
                for (int j : exits) code[j].x = int(code.size());
                break;
            }
            case Node::Group:
                if (n.capture) emit(Inst::Save, 2 * (n.capture - 1));
                gen(*n.kids[0]);
                if (n.runCapture) {  // end slot: filled in after the match from the begin slot (regex_ast.hpp)
                    bool known = false;
                    for (auto& rg : runGroups) {
                        if (rg.first != n.capture - 1) continue;
                        if (!(rg.second == n.set)) throw RegexError("unsupported: one group captures runs of two different sets");
                        known = true;
                    }
                    if (!known) runGroups.emplace_back(n.capture - 1, n.set);
                } else if (n.capture) {
                    emit(Inst::Save, 2 * (n.capture - 1) + 1);
                }
                break;
            case Node::Assert: emit(Inst::Assert, assertIndex(n.look)); break;
            case Node::Atomic: {
                if (atomicCount >= 16000) throw RegexError("unsupported: too many atomic group instances");
                const int g = atomicCount++;  // every expansion copy is its own instance
                emit(Inst::AtomEnter, g);
                gen(*n.kids[0]);
                emit(Inst::AtomExit, g);
                break;
            }
            case Node::Repeat: genRepeat(n); break;
        }
    }

    void genRepeat(const Node& n) {
        const Node& body = *n.kids[0];
        if (n.max < 0) {
            if (nullable(body))
                throw RegexError("unsupported: unbounded repeat of a sub-expression that can match the empty string");
            if (n.min == 0) {
                int sp = emit(Inst::Split);
                int top = int(code.size());
                gen(body);
                emit(Inst::Jump, sp);
                int out = int(code.size());
                code[sp].x = n.greedy ? top : out;
                code[sp].y = n.greedy ? out : top;
            } else {
                for (int i = 0; i + 1 < n.min; ++i) gen(body);
                int top = int(code.size());
                gen(body);
                int sp = emit(Inst::Split);
                int out = int(code.size());
                code[sp].x = n.greedy ? top : out;
                code[sp].y = n.greedy ? out : top;
            }
            return;
        }
        for (int i = 0; i < n.min; ++i) gen(body);
        std::vector<int> splits;
        for (int i = n.min; i < n.max; ++i) {
            int sp = emit(Inst::Split);
            splits.push_back(sp);
            int top = int(code.size());
            if (n.greedy) code[sp].x = top; else code[sp].y = top;
            gen(body);
        }
        int out = int(code.size());
        for (int sp : splits) {
            if (n.greedy) code[sp].y = out; else code[sp].x = out;
        }
    }
};

class PathWalker {
public:
    PathWalker(const std::vector<Inst>& c, bool events) : code(c), recordEvents(events) {}
    std::vector<FollowPath> from(int pc) {
        out.clear();
        steps = 0;
        std::vector<FollowPath::Event> atoms;
        exitVisits = 0;
        walk(pc, TagSet(), 0, atoms, 0);
        return out;
    }

private:
    const std::vector<Inst>& code;
    bool recordEvents;
    std::vector<FollowPath> out;
    size_t steps = 0;
    int exitVisits = 0;

public:
    // walk steps + duplicate checks summed over ALL follow lists of a pattern: the per-list limit alone lets a pattern
    // with thousands of positions buy minutes ("(?>(?>a?){50}){50}")
    static constexpr uint64_t kMaxTotalWork = uint64_t(16) << 20;  // the largest library pattern needs 1.4 M
    uint64_t totalWork = 0;

private:

    void add(int target, TagSet tags, uint32_t cond, const std::vector<FollowPath::Event>& atoms) {
        // a later path to the same target (same atomic history) whose condition set includes an earlier one's can
        // never win
        if ((totalWork += out.size()) > kMaxTotalWork) throw RegexError("unsupported: epsilon closure too large");
        for (const auto& p : out)
            if (p.target == target && (p.cond & ~cond) == 0 && p.atoms == atoms) return;
        if (out.size() >= 4096) throw RegexError("unsupported: too many epsilon paths");
        out.push_back({target, tags, cond, atoms});
    }
    void walk(int pc, TagSet tags, uint32_t cond, std::vector<FollowPath::Event>& atoms, int depth) {
        if (++steps > 2000000 || depth > 100000 || ++totalWork > kMaxTotalWork)
            throw RegexError("unsupported: epsilon closure too large");
        const size_t mark = atoms.size();
        for (;;) {
            const Inst& in = code[pc];
            switch (in.op) {
                case Inst::Char: add(in.x, tags, cond, atoms); atoms.resize(mark); return;
                case Inst::Match: add(kMatchTarget, tags, cond, atoms); atoms.resize(mark); return;
                case Inst::Jump: pc = in.x; break;
                case Inst::Save: tags.set(in.x); ++pc; break;
                case Inst::Assert:
                    cond |= 1u << in.x;
                    if (recordEvents) atoms.push_back({kAssertEvent + in.x, 0});
                    ++pc;
                    break;
                case Inst::AtomEnter: atoms.push_back({in.x + 1, 0}); ++pc; break;
                case Inst::AtomExit: atoms.push_back({-(in.x + 1), ++exitVisits}); ++pc; break;
                case Inst::Split:
                    walk(in.x, tags, cond, atoms, depth + 1);
                    pc = in.y;
                    break;
            }
        }
    }
};

}  // namespace

FollowNfa buildFollowNfa(const ParsedRegex& re) {
    if (re.groupCount > kMaxGpuGroups)
        throw RegexError("unsupported: more than " + std::to_string(kMaxGpuGroups) + " capture groups");
    Builder b;
    b.gen(*re.root);
    b.emit(Inst::Match);

    FollowNfa nfa;
    nfa.groupCount = re.groupCount;
    nfa.groupNames = re.groupNames;
    nfa.positions = b.positions;
    nfa.asserts = b.asserts;
    nfa.atomicCount = b.atomicCount;
    nfa.runGroups = b.runGroups;
    for (size_t i = 0; i < b.asserts.size(); ++i)
        if (b.asserts[i].behind) nfa.behindMask |= 1u << i;
    const int npos = int(b.positions.size());
    nfa.follow.resize(npos + 1);
    PathWalker walker(b.code, b.atomicCount > 0);
    for (int pc = 0; pc < int(b.code.size()); ++pc)
        if (b.code[pc].op == Inst::Char) nfa.follow[b.code[pc].x] = walker.from(pc + 1);
    nfa.follow[npos] = walker.from(0);
    if (getenv("LC_TDFA_WORK_DEBUG")) fprintf(stderr, "follow totalWork %llu\n", (unsigned long long)walker.totalWork);
    for (auto& lst : nfa.follow)
        for (auto& p : lst) nfa.condsUsed |= p.cond;
    return nfa;
}

}  // namespace lcregex
