// follow_nfa.cpp -- AST -> prioritised Thompson program -> epsilon-free follow NFA.
#include "follow_nfa.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>

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
    enum Op : uint8_t { Char, Split, Jump, Save, Assert, AtomEnter, AtomExit, Window, Match } op;
    int x = 0, y = 0;  // Char: position index | Split: preferred, other | Jump: target | Save: slot | Assert: kind | Window: index
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
    // look-ahead windows (regex_ast.hpp Node::window): the next seq.size() bytes must / must not be in seq[0], seq[1], ...
    struct Window {
        std::vector<ByteSet> seq;
        bool negative = false;
    };
    std::vector<Window> windows;

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
            case Node::Assert:
                if (n.window) {
                    if (windows.size() >= 255) throw RegexError("unsupported: too many multi-byte look-aheads");
                    windows.push_back({n.aheadSeq, n.aheadNegative});
                    emit(Inst::Window, int(windows.size()) - 1);
                } else if (!n.aheadSeq.empty() || !n.behindSeq.empty()) {
                    throw RegexError("unsupported: multi-byte look-around left undecided by the parser");
                } else {
                    emit(Inst::Assert, assertIndex(n.look));
                }
                break;
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
        outOpens.clear();
        steps = 0;
        std::vector<FollowPath::Event> atoms;
        std::vector<uint8_t> opens;
        exitVisits = 0;
        walk(pc, TagSet(), 0, atoms, opens, 0);
        return out;
    }
    // the look-ahead windows each path of the list just returned opens (parallel to it), in the order it crosses them
    const std::vector<std::vector<uint8_t>>& opensOfLast() const { return outOpens; }

private:
    const std::vector<Inst>& code;
    bool recordEvents;
    std::vector<FollowPath> out;
    std::vector<std::vector<uint8_t>> outOpens;
    size_t steps = 0;
    int exitVisits = 0;

public:
    // walk steps + duplicate checks summed over ALL follow lists of a pattern: the per-list limit alone lets a pattern
    // with thousands of positions buy minutes ("(?>(?>a?){50}){50}")
    static constexpr uint64_t kMaxTotalWork = uint64_t(16) << 20;  // the largest library pattern needs 1.4 M
    uint64_t totalWork = 0;

private:

    void add(int target, TagSet tags, uint32_t cond, const std::vector<FollowPath::Event>& atoms, const std::vector<uint8_t>& opens) {
        // a later path to the same target (same atomic history, same windows opened) whose condition set includes an earlier
        // one's can never win
        if ((totalWork += out.size()) > kMaxTotalWork) throw RegexError("unsupported: epsilon closure too large");
        for (size_t k = 0; k < out.size(); ++k)
            if (out[k].target == target && (out[k].cond & ~cond) == 0 && out[k].atoms == atoms && outOpens[k] == opens) return;
        if (out.size() >= 4096) throw RegexError("unsupported: too many epsilon paths");
        out.push_back({target, tags, cond, atoms});
        outOpens.push_back(opens);
    }
    void walk(int pc, TagSet tags, uint32_t cond, std::vector<FollowPath::Event>& atoms, std::vector<uint8_t>& opens, int depth) {
        if (++steps > 2000000 || depth > 100000 || ++totalWork > kMaxTotalWork)
            throw RegexError("unsupported: epsilon closure too large");
        const size_t mark = atoms.size(), openMark = opens.size();
        for (;;) {
            const Inst& in = code[pc];
            switch (in.op) {
                case Inst::Char: add(in.x, tags, cond, atoms, opens); atoms.resize(mark); opens.resize(openMark); return;
                case Inst::Match: add(kMatchTarget, tags, cond, atoms, opens); atoms.resize(mark); opens.resize(openMark); return;
                case Inst::Window: opens.push_back(uint8_t(in.x)); ++pc; break;
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
                    walk(in.x, tags, cond, atoms, opens, depth + 1);
                    pc = in.y;
                    break;
            }
        }
    }
};

// ---- multi-byte look-aheads as a product (round 5).  A window w opened at some offset obliges the NEXT k bytes of the input -- whoever
// consumes them: the rest of the pattern, the search wrapper's suffix -- to lie in seq[0..k) (positive), or not all to (negative).
// A thread therefore carries its open obligations: product position = (position p, obligations pending for the byte p consumes).
//   positive (w, i): the byte must be in seq_w[i]; it then becomes (w, i + 1), or is discharged at i + 1 = k.  MATCH (end of input)
//                    with one pending: the window reaches beyond the input -- the path is dead.
//   negative (w, i): the position splits in two by the byte: in seq_w[i] -> (w, i + 1), dead if that completes the window; outside ->
//                    discharged.  Pending at end of input: holds.
// The byte set of a product position is p's set cut by its obligations, so the engines (thread list, tagged DFA, screens) need to
// know nothing about windows: they get an ordinary follow NFA with a few more positions.  Priorities are untouched -- the chain is
// deterministic, the variants of one target are disjoint by byte.  Index layout kept for the callers: base position p keeps index p,
// except the LAST base position (the search wrapper's suffix, regex_handle.cpp), which stays last; product positions lie in between.
void applyWindows(FollowNfa& nfa, const std::vector<std::vector<std::vector<uint8_t>>>& opens, const std::vector<Builder::Window>& windows,
                  bool atomic) {
    if (atomic)
        throw RegexError("unsupported: a multi-byte look-ahead in a pattern with atomic groups (what a failed window gives back is decided "
                         "bytes after the group was left)");
    const int npos = int(nfa.positions.size());
    // obligation entry: window << 16 | progress << 1 | in  (in: this byte is taken inside seq[progress]; always 1 for a positive window)
    using Obl = std::vector<uint32_t>;  // sorted
    struct Prod {
        int p;
        Obl obl;
    };
    std::vector<Prod> prods;           // product positions beyond the base ones: id = npos + index
    std::vector<ByteSet> prodSets;
    std::map<std::pair<int, Obl>, int> ids;
    auto byteSetOf = [&](int p, const Obl& o) {
        ByteSet s = nfa.positions[size_t(p)];
        for (uint32_t e : o) {
            const Builder::Window& w = windows[e >> 16];
            ByteSet c = w.seq[(e >> 1) & 0x7FFFu];
            if (!(e & 1u)) c.invert();
            for (int k = 0; k < 4; ++k) s.w[k] &= c.w[k];
        }
        return s;
    };
    auto idOf = [&](int p, const Obl& o) -> int {  // -1: no byte can be consumed there
        if (o.empty()) return p;
        auto it = ids.find({p, o});
        if (it != ids.end()) return it->second;
        const ByteSet s = byteSetOf(p, o);
        int id = -1;
        if (!s.empty()) {
            if (prods.size() >= size_t(4 * npos + 4096)) throw RegexError("unsupported: multi-byte look-aheads multiply the automaton too far");
            id = npos + int(prods.size());
            prods.push_back({p, o});
            prodSets.push_back(s);
        }
        ids.emplace(std::make_pair(p, o), id);
        return id;
    };
    // what is still pending behind the byte a position with obligations `o` consumes
    auto after = [&](const Obl& o) {
        Obl out;
        for (uint32_t e : o) {
            if (!(e & 1u)) continue;  // negative, byte outside: discharged
            const uint32_t w = e >> 16, i = ((e >> 1) & 0x7FFFu) + 1;
            if (i < windows[w].seq.size()) out.push_back((w << 16) | (i << 1) | 1u);
            // (i == k: a positive window is fulfilled; a negative one would be complete -- such a variant is never created)
        }
        return out;
    };
    // the paths of `base` (with their windows) for a thread whose pending obligations are `pending`: every target in all its variants
    auto expand = [&](const std::vector<FollowPath>& base, const std::vector<std::vector<uint8_t>>& baseOpens, const Obl& pending) {
        std::vector<FollowPath> out;
        for (size_t q = 0; q < base.size(); ++q) {
            Obl o = pending;
            for (uint8_t w : baseOpens[q]) o.push_back((uint32_t(w) << 16) | 1u);
            std::sort(o.begin(), o.end());
            o.erase(std::unique(o.begin(), o.end()), o.end());
            if (base[q].target == kMatchTarget) {
                bool positivePending = false;
                for (uint32_t e : o) positivePending = positivePending || !windows[e >> 16].negative;
                if (!positivePending) out.push_back(base[q]);
                continue;
            }
            std::vector<size_t> neg;  // entries that split the target by the byte
            for (size_t k = 0; k < o.size(); ++k)
                if (windows[o[k] >> 16].negative) neg.push_back(k);
            if (neg.size() > 6) throw RegexError("unsupported: too many overlapping negative multi-byte look-aheads");
            for (uint32_t choice = 0; choice < (1u << neg.size()); ++choice) {
                Obl v = o;
                bool dead = false;
                for (size_t k = 0; k < neg.size(); ++k) {
                    const bool in = (choice >> k) & 1u;
                    uint32_t& e = v[neg[k]];
                    e = (e & ~1u) | (in ? 1u : 0u);
                    if (in && ((e >> 1) & 0x7FFFu) + 1 == windows[e >> 16].seq.size()) dead = true;  // the forbidden text would be complete
                }
                if (dead) continue;
                std::sort(v.begin(), v.end());
                const int id = idOf(base[q].target, v);
                if (id < 0) continue;
                FollowPath p = base[q];
                p.target = id;
                bool dup = false;
                for (const auto& e : out) dup = dup || (e.target == p.target && (e.cond & ~p.cond) == 0);
                if (!dup) out.push_back(std::move(p));
            }
        }
        return out;
    };
    std::vector<std::vector<FollowPath>> follow(size_t(npos) + 1);
    for (int p = 0; p <= npos; ++p) follow[size_t(p)] = expand(nfa.follow[size_t(p)], opens[size_t(p)], Obl());
    std::vector<std::vector<FollowPath>> prodFollow;
    for (size_t k = 0; k < prods.size(); ++k) {  // (prods grows while we go)
        const Prod pr = prods[k];
        prodFollow.push_back(expand(nfa.follow[size_t(pr.p)], opens[size_t(pr.p)], after(pr.obl)));
    }
    // final numbering: base 0 .. npos-2 | product positions | base npos-1 | (start pseudo-position)
    const int total = npos + int(prods.size());
    auto renum = [&](int id) { return id < 0 ? id : id == npos - 1 ? total - 1 : id >= npos ? id - 1 : id; };
    std::vector<ByteSet> positions;
    positions.resize(size_t(total));
    std::vector<std::vector<FollowPath>> all(size_t(total) + 1);
    for (int p = 0; p < npos; ++p) {
        positions[size_t(renum(p))] = nfa.positions[size_t(p)];
        all[size_t(renum(p))] = std::move(follow[size_t(p)]);
    }
    for (size_t k = 0; k < prods.size(); ++k) {
        positions[size_t(renum(npos + int(k)))] = prodSets[k];
        all[size_t(renum(npos + int(k)))] = std::move(prodFollow[k]);
    }
    all[size_t(total)] = std::move(follow[size_t(npos)]);
    for (auto& lst : all)
        for (auto& p : lst) p.target = renum(p.target);
    nfa.positions.swap(positions);
    nfa.follow.swap(all);
}

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
    std::vector<std::vector<std::vector<uint8_t>>> opens(size_t(npos) + 1);  // per follow list, per path: the windows it opens
    PathWalker walker(b.code, b.atomicCount > 0);
    for (int pc = 0; pc < int(b.code.size()); ++pc)
        if (b.code[pc].op == Inst::Char) {
            nfa.follow[b.code[pc].x] = walker.from(pc + 1);
            opens[size_t(b.code[pc].x)] = walker.opensOfLast();
        }
    nfa.follow[npos] = walker.from(0);
    opens[size_t(npos)] = walker.opensOfLast();
    if (!b.windows.empty()) applyWindows(nfa, opens, b.windows, b.atomicCount > 0);
    if (getenv("LC_TDFA_WORK_DEBUG")) fprintf(stderr, "follow totalWork %llu\n", (unsigned long long)walker.totalWork);
    for (auto& lst : nfa.follow)
        for (auto& p : lst) nfa.condsUsed |= p.cond;
    return nfa;
}

}  // namespace lcregex
