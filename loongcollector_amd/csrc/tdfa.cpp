// tdfa.cpp -- follow NFA -> tagged DFA tables.  See tdfa.hpp.
#include "tdfa.hpp"

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <string>
#include <unordered_map>

namespace lcregex {

namespace {

constexpr uint8_t kRegTmp = 0xFD;  // placeholder, patched to the real scratch register at the end

// Atomic groups.  Every thread that entered atomic-group instance g through the same source thread at the same step
// belongs to one SEGMENT; the first (highest-priority) member that leaves g commits the group for that entry: every
// lower-priority member -- still inside g, or one that left earlier -- must die ("the first way the body matches is
// final").  An item carries its unsettled memberships; one is dropped once no higher-priority member is still inside.
struct LinEntry {
    int g, seg;
    bool exited;
    bool operator==(const LinEntry& o) const { return g == o.g && seg == o.seg && exited == o.exited; }
};
struct Item {
    int pos;
    std::vector<int> regs;      // per slot: register id, or -1
    std::vector<LinEntry> lin;  // unsettled atomic-segment memberships
};
struct Cand {
    int pos, src;
    TagSet tags;
    std::vector<LinEntry> lin;                        // lineage of the source item (updated by commitAtomic)
    const std::vector<FollowPath::Event>* events;     // the path's events (nullptr: pattern has no atomic group)
    bool targetOk;                                    // ends on a position that accepts the byte (or on MATCH at END)
    uint32_t cond;
};

// Priority-ordered commit over ALL epsilon paths of a step, viable or not.
//   * Leaving group g commits the group for that entry: the segment is CLOSED, by (source item, exit visit), for every
//     lower-priority thread -- also when the leaving thread then finds no byte it can consume: the body HAS matched,
//     which is all an atomic group asks.
//   * A path that is a member of a closed segment is dead, unless it leaves the group through the very same exit visit
//     of the same source item (then it IS the committed body match and merely continues differently after the group).
//     A dead path contributes nothing.
//   * A failing assertion stops a live path there; what it did before still happened.
// Survivors with a consumable target are kept; memberships nobody can act on any more are dropped and indistinguishable
// survivors collapse onto the first.
// pair tests of the redundancy elimination below, summed over one buildTdfa() call (reset there): the elimination is
// cubic in the number of survivors of a step, and a configuration file must not be able to buy minutes of Init
thread_local uint64_t tlsCommitWork = 0;
thread_local uint64_t kMaxCommitWork = 20u << 20;  // TdfaLimits::maxCommitWork of the construction in progress

std::vector<Cand> commitAtomic(std::vector<Cand> cands, uint32_t holds) {
    struct Closure {
        int g, seg, src, visit;
    };
    std::vector<Cand> kept;
    std::vector<Closure> closed;
    std::map<std::pair<int, int>, int> freshSeg;
    auto closureOf = [&](int g, int seg) -> const Closure* {
        for (const auto& c : closed)
            if (c.g == g && c.seg == seg) return &c;
        return nullptr;
    };
    for (auto& c : cands) {
        if (!c.events) {  // no atomic groups anywhere in the pattern
            if (c.targetOk && !(c.cond & ~holds)) kept.push_back(std::move(c));
            continue;
        }
        const auto& ev = *c.events;
        // exit visit through which this path leaves a membership it holds at path start (or 0)
        auto exitVisitFor = [&](int g, size_t fromEvent) {
            int depth = 0;  // re-entries of the same instance cannot nest, but be safe
            for (size_t i = fromEvent; i < ev.size(); ++i) {
                if (ev[i].code >= kAssertEvent) {
                    if (!((holds >> (ev[i].code - kAssertEvent)) & 1u)) return 0;  // the path stops before leaving
                } else if (ev[i].code == g + 1) {
                    ++depth;
                } else if (ev[i].code == -(g + 1)) {
                    if (depth == 0) return int(ev[i].visit);
                    --depth;
                }
            }
            return 0;
        };
        bool dead = false;
        for (const auto& e : c.lin) {
            const Closure* cl = closureOf(e.g, e.seg);
            if (!cl) continue;
            if (e.exited || cl->src != c.src || exitVisitFor(e.g, 0) != cl->visit) dead = true;
        }
        if (dead) continue;
        bool assertsOk = true;
        for (size_t i = 0; i < ev.size() && !dead; ++i) {
            const int code = ev[i].code;
            if (code >= kAssertEvent) {
                if (!((holds >> (code - kAssertEvent)) & 1u)) {
                    assertsOk = false;
                    break;
                }
            } else if (code > 0) {
                const int g = code - 1;
                auto it = freshSeg.find({c.src, g});
                if (it == freshSeg.end())
                    it = freshSeg.emplace(std::make_pair(c.src, g), 1000000 + int(freshSeg.size())).first;
                const Closure* cl = closureOf(g, it->second);
                if (cl && (cl->src != c.src || exitVisitFor(g, i + 1) != cl->visit)) {
                    dead = true;
                    break;
                }
                c.lin.push_back({g, it->second, false});
            } else {
                const int g = -code - 1;
                for (size_t k = c.lin.size(); k-- > 0;)
                    if (c.lin[k].g == g && !c.lin[k].exited) {
                        c.lin[k].exited = true;
                        if (!closureOf(g, c.lin[k].seg)) closed.push_back({g, c.lin[k].seg, c.src, int(ev[i].visit)});
                        break;
                    }
            }
        }
        if (dead || !assertsOk || !c.targetOk) continue;
        kept.push_back(std::move(c));
    }
    for (size_t i = 0; i < kept.size(); ++i) {
        auto& lin = kept[i].lin;
        for (size_t k = 0; k < lin.size();) {
            bool contested = !lin[k].exited;  // still inside: always kept
            if (lin[k].exited) {
                for (size_t j = 0; j < i && !contested; ++j)
                    for (const auto& e : kept[j].lin)
                        if (e.g == lin[k].g && e.seg == lin[k].seg && !e.exited) contested = true;
                if ((tlsCommitWork += i) > kMaxCommitWork)
                    throw RegexError("tdfa: construction work limit (atomic groups: too many concurrent alternatives)");
            }
            if (contested) ++k;
            else lin.erase(lin.begin() + long(k));
        }
    }
    // A lower-priority survivor `lo` on the same position as a higher one `hi` is redundant when
    //   (1) whatever can kill hi's line also kills lo's: each membership of hi is shared by lo, or no survivor above hi
    //       is still inside that segment (then only hi's own descendants can ever close it, which lo's line mirrors);
    //   (2) whatever lo's line could kill, hi's line kills too: each group lo is still inside is the very segment hi is
    //       inside, or no survivor below lo belongs to it;
    // lo then mirrors hi move for move at lower priority, can never outlive it and never changes anybody's fate.
    // (Dropping one survivor can make another one redundant, so repeat until nothing changes.)
    std::vector<Cand> out;
    {  // survivors without any membership: the first one on a position makes every later one redundant -- settled in one
       // linear pass, the cubic machinery below only sees what is left
        std::map<int, bool> plainSeen;
        out.reserve(kept.size());
        for (auto& c : kept) {
            if (c.lin.empty()) {
                auto ins = plainSeen.emplace(c.pos, true);
                if (!ins.second) continue;
            }
            out.push_back(std::move(c));
        }
    }
    {  // Nobody holds a membership (a pattern whose events are assertions only, or a step outside every atomic group): positions
       // are unique after the pass above, `mirrors` needs two survivors on one position and `familyMirror` a segment to rename --
       // there is nothing for the pair loop to find, and on an automaton of tens of thousands of states its n^2 tests per step were
       // what exhausted maxCommitWork ("too many concurrent alternatives" for SYSLOGLINE, which has no atomic group left at all).
        bool anyMembership = false;
        for (const auto& c : out) anyMembership |= !c.lin.empty();
        if (!anyMembership) return out;
    }
    auto holds_ = [](const Cand& c, const LinEntry& e, bool insideOnly) {
        ++tlsCommitWork;
        for (const auto& x : c.lin)
            if (x.g == e.g && x.seg == e.seg && (!insideOnly || !x.exited)) return true;
        return false;
    };
    auto mirrors = [&](size_t hi, size_t lo) {
        if (out[hi].pos != out[lo].pos) return false;
        for (const auto& e : out[hi].lin) {
            if (holds_(out[lo], e, false)) continue;
            for (size_t k = 0; k < hi; ++k)
                if (holds_(out[k], e, true)) return false;
        }
        for (const auto& e : out[lo].lin) {
            if (e.exited || holds_(out[hi], e, true)) continue;
            for (size_t k = lo + 1; k < out.size(); ++k)
                if (holds_(out[k], e, false)) return false;
        }
        return true;
    };
    // The same argument for whole families: if renaming segments (sigma) maps the survivors holding sigma's domain one
    // to one, in priority order, onto the survivors holding its range -- same positions, same memberships after
    // renaming, every image above its original -- the domain family replays the range family at lower priority.
    auto familyMirror = [&](size_t hi, size_t lo, std::vector<size_t>& drop) {
        const auto &H = out[hi].lin, &L = out[lo].lin;
        if (out[hi].pos != out[lo].pos || H.size() != L.size()) return false;
        std::vector<std::pair<LinEntry, LinEntry>> sigma;  // (from, to), compared on (g, seg)
        auto same = [](const LinEntry& a, const LinEntry& b) { return a.g == b.g && a.seg == b.seg; };
        for (size_t k = 0; k < H.size(); ++k) {
            if (H[k].g != L[k].g || H[k].exited != L[k].exited) return false;
            if (H[k].seg != L[k].seg) sigma.emplace_back(L[k], H[k]);
        }
        if (sigma.empty()) return false;
        for (const auto& a : sigma)
            for (const auto& b : sigma)
                if (same(a.first, b.second) || (same(a.first, b.first) != same(a.second, b.second))) return false;
        std::vector<size_t> dom, ran;
        tlsCommitWork += out.size();
        for (size_t i = 0; i < out.size(); ++i) {
            bool inDom = false, inRan = false;
            for (const auto& e : out[i].lin)
                for (const auto& m : sigma) {
                    inDom |= same(e, m.first);
                    inRan |= same(e, m.second);
                }
            if (inDom && inRan) return false;
            if (inDom) dom.push_back(i);
            if (inRan) ran.push_back(i);
        }
        if (dom.size() != ran.size()) return false;
        for (size_t t = 0; t < dom.size(); ++t) {
            const Cand &d = out[dom[t]], &r = out[ran[t]];
            if (ran[t] >= dom[t] || d.pos != r.pos || d.lin.size() != r.lin.size()) return false;
            for (size_t k = 0; k < d.lin.size(); ++k) {
                LinEntry e = d.lin[k];
                for (const auto& m : sigma)
                    if (same(e, m.first)) {
                        e.seg = m.second.seg;
                        break;
                    }
                if (!(e == r.lin[k])) return false;
            }
        }
        drop = dom;
        return true;
    };
    for (bool changed = true; changed;) {
        changed = false;
        for (size_t i = 1; i < out.size() && !changed; ++i)
            for (size_t j = 0; j < i && !changed; ++j) {
                if (++tlsCommitWork > kMaxCommitWork)
                    throw RegexError("tdfa: construction work limit (atomic groups: too many concurrent alternatives)");
                std::vector<size_t> drop;
                if (mirrors(j, i)) {
                    out.erase(out.begin() + long(i));
                    changed = true;
                } else if (familyMirror(j, i, drop)) {
                    for (size_t t = drop.size(); t-- > 0;) out.erase(out.begin() + long(drop[t]));
                    changed = true;
                }
            }
    }
    return out;
}

struct State {
    std::vector<Item> items;
    uint32_t prevCtx = 0;  // behind-assertions that hold at this state's offset (masked to the ones still observable)
    int nregs = 0;
};

std::string keyOf(const State& s) {
    std::string k;
    k.append(reinterpret_cast<const char*>(&s.prevCtx), sizeof s.prevCtx);
    for (const auto& it : s.items) {
        k.append(reinterpret_cast<const char*>(&it.pos), sizeof(int));
        for (int r : it.regs) {
            int16_t v = int16_t(r);
            k.append(reinterpret_cast<const char*>(&v), sizeof v);
        }
        for (const auto& e : it.lin) {
            const int32_t v[3] = {e.g, e.seg, e.exited ? 1 : 0};
            k.append(reinterpret_cast<const char*>(v), sizeof v);
        }
        k.push_back('|');
    }
    return k;
}

// order a set of injective register moves (dst <- src) so that no source is clobbered before it is read
std::vector<uint16_t> scheduleMoves(std::vector<std::pair<int, int>> regMoves, const std::vector<int>& posDsts) {
    std::vector<uint16_t> out;
    auto push = [&](int dst, int src) { out.push_back(uint16_t(dst | (src << 8))); };
    while (!regMoves.empty()) {
        bool progressed = false;
        for (size_t i = 0; i < regMoves.size(); ++i) {
            int d = regMoves[i].first;
            bool blocked = false;
            for (size_t j = 0; j < regMoves.size(); ++j)
                if (j != i && regMoves[j].second == d) {
                    blocked = true;
                    break;
                }
            if (!blocked) {
                push(d, regMoves[i].second);
                regMoves.erase(regMoves.begin() + long(i));
                progressed = true;
                break;
            }
        }
        if (progressed) continue;
        // only cycles remain: park one destination's old value in the scratch register
        int d = regMoves[0].first;
        push(kRegTmp, d);
        for (auto& m : regMoves)
            if (m.second == d) m.second = kRegTmp;
    }
    for (int d : posDsts) push(d, kRegPos);
    return out;
}

}  // namespace

// ---- dead stores.  A search pattern that ends in a greedy field -- "(?s:.*?)(... (.*))(?s:.*)", every Grok format with a
// GREEDYDATA tail -- re-derives, at EVERY byte of the tail, the thread that would take over if the field ended here, and with it
// the stamps of "the field ends here, the match ends here": a register program on 90-98 % of the bytes of such lines (measured on the
// configs[2] corpus), two stamps each, which the kernels pay for byte by byte.  Nearly all of them are dead: the byte that really
// ends the field stamps the same registers again before anything reads them, and at the end of the line the field's own thread
// wins, whose final map reads "end of line", not those registers.  Classic liveness over the automaton: a register is live in a
// state if some path from it reads the register (a copy's source, or the final map of a state the line can end in) before writing
// it; a store to a register that is not live behind the transition is dropped.  Semantics unchanged, tables smaller, and the
// self-loop of the tail carries no program at all.
static void eliminateDeadStores(TdfaTables& T) {
    static const bool off = getenv("LC_TDFA_NO_DSE") != nullptr;  // (A/B measurements)
    if (off || T.ops.empty()) return;
    const uint32_t ncls = T.nClasses, nStates = T.nStates;
    const size_t nLists = T.opsStart.size() - 1;
    typedef std::array<uint64_t, 4> RegSet;  // 256 registers
    auto has = [](const RegSet& s, unsigned r) { return (s[r >> 6] >> (r & 63)) & 1; };
    auto add = [](RegSet& s, unsigned r) { s[r >> 6] |= uint64_t(1) << (r & 63); };
    auto del = [](RegSet& s, unsigned r) { s[r >> 6] &= ~(uint64_t(1) << (r & 63)); };
    std::vector<std::vector<uint16_t>> lists(nLists);
    for (size_t id = 1; id < nLists; ++id) {
        const uint32_t at = T.opsStart[id];
        if (at >= T.ops.size() || T.opsStart[id + 1] == at) continue;
        const uint32_t n = T.ops[at];
        lists[id].assign(T.ops.begin() + at + 1, T.ops.begin() + at + 1 + n);
    }
    // what a line that ends in state s reads
    std::vector<RegSet> live(nStates, RegSet{{0, 0, 0, 0}});
    const size_t slots = T.nSlots ? T.nSlots : 1;
    for (uint32_t s = 1; s < nStates; ++s) {
        if (T.finalId[s] == 0xFFFF) continue;
        for (size_t k = 0; k < slots; ++k) {
            const uint8_t m = T.finalMap[size_t(T.finalId[s]) * slots + k];
            if (m != kRegPos && m != kRegNone) add(live[s], m);
        }
    }
    // live-in of the source state through one transition: the ops run in order, so walk them backwards
    auto through = [&](const std::vector<uint16_t>& ops, RegSet after) {
        for (size_t k = ops.size(); k-- > 0;) {
            const unsigned dst = ops[k] & 0xFF, src = ops[k] >> 8;
            if (!has(after, dst)) continue;  // a dead store reads nothing
            del(after, dst);
            if (src != kRegPos) add(after, src);
        }
        return after;
    };
    for (bool changed = true; changed;) {
        changed = false;
        for (uint32_t s = nStates; s-- > 1;) {
            RegSet acc = live[s];
            for (uint32_t c = 0; c < ncls; ++c) {
                const uint32_t e = T.trans[size_t(s) * ncls + c];
                const uint32_t t = e & 0xFFFF;
                if (!t) continue;
                const RegSet in = through(lists[e >> 16], live[t]);
                for (int w = 0; w < 4; ++w) acc[w] |= in[w];
            }
            if (acc != live[s]) {
                live[s] = acc;
                changed = true;
            }
        }
    }
    // rewrite: per transition, keep the stores that are live behind it
    std::map<std::vector<uint16_t>, uint32_t> interned;
    std::vector<std::vector<uint16_t>> newLists(1);
    interned.emplace(std::vector<uint16_t>(), 0u);
    size_t before = 0, after = 0;
    for (uint32_t s = 1; s < nStates; ++s)
        for (uint32_t c = 0; c < ncls; ++c) {
            uint32_t& e = T.trans[size_t(s) * ncls + c];
            const uint32_t t = e & 0xFFFF;
            const std::vector<uint16_t>& ops = lists[e >> 16];
            if (ops.empty()) continue;
            std::vector<uint16_t> kept;
            if (t) {
                RegSet need = live[t];
                std::vector<char> keep(ops.size(), 0);
                for (size_t k = ops.size(); k-- > 0;) {
                    const unsigned dst = ops[k] & 0xFF, src = ops[k] >> 8;
                    if (!has(need, dst)) continue;
                    keep[k] = 1;
                    del(need, dst);
                    if (src != kRegPos) add(need, src);
                }
                for (size_t k = 0; k < ops.size(); ++k)
                    if (keep[k]) kept.push_back(ops[k]);
            }
            before += ops.size();
            after += kept.size();
            auto it = interned.find(kept);
            if (it == interned.end()) {
                it = interned.emplace(kept, uint32_t(newLists.size())).first;
                newLists.push_back(kept);
            }
            e = t | (it->second << 16);
        }
    // (always rebuilt: the transitions above now carry the ids of newLists, whatever was or was not dropped)
    (void)before;
    (void)after;
    T.opsStart.assign(1, 0);
    T.ops.clear();
    for (const auto& l : newLists) {
        if (!l.empty()) {
            T.ops.push_back(uint16_t(l.size()));
            T.ops.insert(T.ops.end(), l.begin(), l.end());
        }
        T.opsStart.push_back(uint32_t(T.ops.size()));
    }
}

// ---- state minimisation.  The subset construction tells states apart that behave alike: the same accepting row, and on every byte
// class the same register program into states that again behave alike.  Moore's refinement finds the coarsest such partition and
// one state per block is kept; register programs and final maps are compared by CONTENT, register names included, so nothing
// about the captures changes.  Search patterns shrink most -- after the dead-store pass their tail states differ in nothing that
// is ever read: CATALINALOG 1 410 -> 136 states (back inside the LDS window), TOMCATLOG 1 602 -> 338, the anchored CRONLOG
// 2 223 -> 898; the headline regexes are minimal as built.
void minimizeTdfaStates(TdfaTables& T) {
    static const bool off = getenv("LC_TDFA_NO_MINIMIZE") != nullptr;  // (A/B measurements)
    const uint32_t n = T.nStates, ncls = T.nClasses;
    if (off || n <= 2) return;
    // register programs by content
    const size_t nLists = T.opsStart.size() - 1;
    std::vector<uint32_t> canon(nLists, 0);
    {
        std::map<std::vector<uint16_t>, uint32_t> byContent;
        byContent.emplace(std::vector<uint16_t>(), 0u);
        for (size_t id = 1; id < nLists; ++id) {
            const uint32_t at = T.opsStart[id];
            std::vector<uint16_t> ops;
            if (at < T.ops.size() && T.opsStart[id + 1] != at) ops.assign(T.ops.begin() + at + 1, T.ops.begin() + at + 1 + T.ops[at]);
            canon[id] = byContent.emplace(ops, uint32_t(byContent.size())).first->second;
        }
    }
    std::vector<uint32_t> block(n), next(n);
    uint32_t nBlocks = 0;
    {
        std::map<uint32_t, uint32_t> byFinal;  // (final rows are interned by content when they are built)
        for (uint32_t s = 0; s < n; ++s) block[s] = byFinal.emplace(uint32_t(T.finalId[s]), uint32_t(byFinal.size())).first->second;
        nBlocks = uint32_t(byFinal.size());
    }
    struct VecHash {
        size_t operator()(const std::vector<uint32_t>& v) const {
            uint64_t h = 1469598103934665603ull;
            for (uint32_t x : v) h = (h ^ x) * 1099511628211ull;
            return size_t(h);
        }
    };
    std::vector<uint32_t> sig(1 + 2 * size_t(ncls));
    // Moore's refinement needs as many rounds as the longest string that tells two states apart: tens for a log format, but n for a
    // chain ("(?i)" + 5 000 letters: n^2 work).  A pattern comes from a configuration file and Init must answer quickly: past 400
    // rounds or this much work the automaton simply stays as it was built.
    uint64_t budget = uint64_t(1) << 30;
    for (int rounds = 0;; ++rounds) {
        const uint64_t round = uint64_t(n) * (1 + 2 * uint64_t(ncls));
        if (round > budget || rounds >= 400) return;
        budget -= round;
        std::unordered_map<std::vector<uint32_t>, uint32_t, VecHash> ids;
        ids.reserve(size_t(nBlocks) * 2);
        for (uint32_t s = 0; s < n; ++s) {
            sig[0] = block[s];
            for (uint32_t c = 0; c < ncls; ++c) {
                const uint32_t e = T.trans[size_t(s) * ncls + c];
                sig[1 + 2 * c] = block[e & 0xFFFF];
                sig[2 + 2 * c] = canon[e >> 16];
            }
            next[s] = ids.emplace(sig, uint32_t(ids.size())).first->second;
        }
        const uint32_t count = uint32_t(ids.size());
        block.swap(next);
        if (count == nBlocks) break;
        nBlocks = count;
    }
    if (nBlocks == n) return;
    // one state per block, in order of first appearance (the dead state stays state 0)
    std::vector<uint32_t> newId(nBlocks, 0xFFFFFFFFu), rep;
    for (uint32_t s = 0; s < n; ++s)
        if (newId[block[s]] == 0xFFFFFFFFu) {
            newId[block[s]] = uint32_t(rep.size());
            rep.push_back(s);
        }
    auto map = [&](uint32_t s) { return newId[block[s]]; };
    std::vector<uint32_t> trans(rep.size() * size_t(ncls));
    std::vector<uint16_t> finalId(rep.size());
    for (size_t k = 0; k < rep.size(); ++k) {
        finalId[k] = T.finalId[rep[k]];
        for (uint32_t c = 0; c < ncls; ++c) {
            const uint32_t e = T.trans[size_t(rep[k]) * ncls + c];
            trans[k * ncls + c] = map(e & 0xFFFF) | (e & 0xFFFF0000u);
        }
    }
    T.startState = map(T.startState);
    if (T.missState) T.missState = map(T.missState);
    for (auto& st : T.startAfter) st = map(st);
    T.trans.swap(trans);
    T.finalId.swap(finalId);
    T.nStates = uint32_t(rep.size());
}

static TdfaTables buildTdfaImpl(const FollowNfa& nfa, const TdfaLimits& limits, const TdfaLazyGuide* guide, TdfaLazyReport* report);

TdfaTables buildTdfaUncached(const FollowNfa& nfa, const TdfaLimits& limits) { return buildTdfaImpl(nfa, limits, nullptr, nullptr); }

// The LAZY / data-guided form (round 6).  The 36 search forms and 9 anchored forms of BASELINE configs[2] that stay on the thread-list
// engine need 170 000 to more than 600 000 states when every (state, byte class) is explored -- but real log lines visit a few thousand
// of them.  Here the same construction (same steps, same register programs, same dead-store and minimisation passes) computes only the
// transitions that the SAMPLE values take, value by value from the start state; every other transition leads to the MISS state, a
// sink the kernels report (csrc/tdfa_l2_kernel.hpp: the value is then walked by the thread-list kernels from its first byte, as
// before).  On every path it knows the partial automaton IS the full one, so a value it decides is decided as the full tagged DFA --
// and as the thread-list engine built from the same follow NFA -- would.  Nothing is cached on disk: the tables depend on the sample.
TdfaTables buildTdfaLazy(const FollowNfa& nfa, const TdfaLimits& limits, const TdfaLazyGuide& guide, TdfaLazyReport* report) {
    return buildTdfaImpl(nfa, limits, &guide, report);
}

static TdfaTables buildTdfaImpl(const FollowNfa& nfa, const TdfaLimits& limits, const TdfaLazyGuide* guide, TdfaLazyReport* report) {
    const int npos = int(nfa.positions.size());
    const int nslots = nfa.slotCount();
    TdfaTables T;
    T.nSlots = uint32_t(nslots);

    // ---- byte classes: bytes that no position set and no assertion can tell apart
    T.classMap.assign(256, 0);
    std::vector<unsigned> classRep;
    {
        std::map<std::vector<bool>, int> sig2cls;
        for (unsigned b = 0; b < 256; ++b) {
            std::vector<bool> sig;
            sig.reserve(size_t(npos) + 4);
            for (int p = 0; p < npos; ++p) sig.push_back(nfa.positions[p].has(b));
            for (const auto& a : nfa.asserts) sig.push_back(a.set.has(b));
            auto it = sig2cls.find(sig);
            if (it == sig2cls.end()) {
                it = sig2cls.emplace(sig, int(classRep.size())).first;
                classRep.push_back(b);
            }
            T.classMap[b] = uint8_t(it->second);
        }
    }
    const int ncls = int(classRep.size());
    T.nClasses = uint32_t(ncls);

    // which prev-byte facts each position's continuation can observe
    std::vector<uint32_t> need(size_t(npos) + 1, 0);
    for (int p = 0; p <= npos; ++p)
        for (const auto& path : nfa.follow[p]) need[p] |= path.cond & nfa.behindMask;

    // The device format addresses transition rows with 16 bits (device_tables.h): states x (classes + 1) x 4 bytes must
    // stay under 64 KiB - 320.  Knowing that up front makes hopeless determinisations (Grok log formats) fail fast.
    const uint32_t maxStates =
        limits.ldsWindow ? std::min<uint32_t>(limits.maxStates, (65536u - 320u) / (uint32_t(ncls + 1) * 4u)) : limits.maxStates;
    std::vector<State> states;
    // (keys are tens of kilobytes for the automata that take long: eight bytes per multiply instead of std::hash's one per step.
    // The map is only ever looked up, never walked: the numbering of the states does not depend on the hash.)
    struct KeyHash {
        size_t operator()(const std::string& k) const {
            uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t(k.size()) * 0xD6E8FEB86659FD93ull);
            const char* p = k.data();
            size_t n = k.size();
            for (; n >= 8; n -= 8, p += 8) {
                uint64_t w;
                std::memcpy(&w, p, 8);
                h = (h ^ w) * 0xD6E8FEB86659FD93ull;
                h ^= h >> 29;
            }
            uint64_t w = 0;
            if (n) std::memcpy(&w, p, n);
            h = (h ^ w) * 0xD6E8FEB86659FD93ull;
            h ^= h >> 33;  // (murmur3's finaliser: the buckets are taken modulo a prime, but equal low halves must not cluster)
            h *= 0xFF51AFD7ED558CCDull;
            h ^= h >> 33;
            h *= 0xC4CEB9FE1A85EC53ull;
            return size_t(h ^ (h >> 33));
        }
    };
    std::unordered_map<std::string, uint32_t, KeyHash> index;
    std::deque<uint32_t> work;
    states.emplace_back();  // state 0 = dead
    auto intern = [&](State&& s) -> uint32_t {
        std::string k = keyOf(s);
        auto it = index.find(k);
        if (it != index.end()) return it->second;
        if (states.size() >= maxStates) throw RegexError("tdfa: state limit exceeded");
        uint32_t id = uint32_t(states.size());
        states.push_back(std::move(s));
        index.emplace(std::move(k), id);
        work.push_back(id);
        return id;
    };

    {
        State s0;
        Item it;
        it.pos = nfa.startIndex();
        it.regs.assign(size_t(nslots), -1);
        s0.items.push_back(std::move(it));
        s0.prevCtx = nfa.behindBits(kEdge) & need[nfa.startIndex()];
        T.startState = intern(std::move(s0));
    }
    if (nfa.searchPrefix >= 0) {
        for (int c = 0; c < ncls; ++c) {
            State s;
            Item it;
            it.pos = nfa.searchPrefix;
            it.regs.assign(size_t(nslots), -1);
            s.items.push_back(std::move(it));
            s.prevCtx = nfa.behindBits(int(classRep[size_t(c)])) & need[size_t(nfa.searchPrefix)];
            T.startAfter.push_back(intern(std::move(s)));
        }
    }

    std::map<std::vector<uint16_t>, uint32_t> opListIds;
    std::vector<std::vector<uint16_t>> opLists(1);  // id 0 = empty
    std::vector<std::vector<uint32_t>> transRows;   // filled per state id
    transRows.emplace_back(size_t(ncls), 0u);       // dead row
    int maxRegs = 0;
    bool usedTmp = false;
    tlsCommitWork = 0;
    kMaxCommitWork = limits.maxCommitWork;
    uint64_t pathWork = 0;  // epsilon paths looked at so far (a config-supplied pattern must not buy minutes of Init)
    std::vector<uint32_t> targetSeen(nfa.positions.size(), 0u);
    uint32_t seenStamp = 0;

    struct WorkReport {
        const uint64_t& w;
        ~WorkReport() {
            if (getenv("LC_TDFA_WORK_DEBUG"))
                fprintf(stderr, "tdfa pathWork %llu commitWork %llu\n", (unsigned long long)w, (unsigned long long)tlsCommitWork);
        }
    } workReport{pathWork};
    // ---- Patterns without atomic groups (every one whose groups the elision pass removed, and most others): the step of one
    // (state, byte class) without the general machinery.  No memberships: the survivors are the first path per target position, in
    // priority order, that consumes the byte and whose assertions hold; nothing to commit, nothing to collapse.  Same states, same
    // numbering, same register programs as the general loop below (which patterns with atomic groups still take) -- 98 % of the
    // steps of a large automaton land on a state that exists already, and this path finds that out from a key assembled in a reused
    // buffer, without building the State: no map, no vector, no allocation per step.  (An anchored Grok format of 30 000 states:
    // 10 s -> 2.6 s of construction.)
    const bool atomicPattern = nfa.atomicCount > 0;
    std::vector<uint32_t> viableStart, viablePath;  // per (position, class): the follow paths whose target takes the class
    if (!atomicPattern) {
        viableStart.reserve((size_t(npos) + 1) * size_t(ncls) + 1);
        for (int p = 0; p <= npos; ++p)
            for (int c = 0; c < ncls; ++c) {
                viableStart.push_back(uint32_t(viablePath.size()));
                const auto& lst = nfa.follow[size_t(p)];
                for (size_t i = 0; i < lst.size(); ++i)
                    if (lst[i].target >= 0 && nfa.positions[size_t(lst[i].target)].has(classRep[size_t(c)])) viablePath.push_back(uint32_t(i));
            }
        viableStart.push_back(uint32_t(viablePath.size()));
    }
    struct Survivor {
        int pos, src;
        const TagSet* tags;
        const std::vector<LinEntry>* lin;  // unsettled memberships (patterns with atomic groups), or nullptr
    };
    std::vector<Survivor> survivors;
    std::vector<int> newRegs;                       // survivors x slots
    std::vector<int> oldToNew(size_t(kMaxTdfaRegs) + 8, -1), freshToNew(size_t(nslots) + 1, -1);
    std::vector<int> oldTouched, freshTouched;
    std::vector<std::pair<int, int>> regMovesBuf;
    std::vector<int> posDstsBuf;
    std::string keyBuf;
    // the step's target state and register program from `survivors` (priority order) -> the transition word
    std::vector<int> segNames;
    auto emitTransition = [&](uint32_t sid, unsigned b) -> uint32_t {
        // registers of the target state: canonical names by first appearance (items in order, slots in order)
        for (int r : oldTouched) oldToNew[size_t(r)] = -1;
        for (int sl : freshTouched) freshToNew[size_t(sl)] = -1;
        oldTouched.clear();
        freshTouched.clear();
        int nNew = 0;
        uint32_t needMask = 0;
        newRegs.resize(survivors.size() * size_t(nslots));
        {
            const State& S = states[sid];
            for (size_t i = 0; i < survivors.size(); ++i) {
                const Survivor& n = survivors[i];
                const std::vector<int>& srcRegs = S.items[size_t(n.src)].regs;
                int* out = newRegs.data() + i * size_t(nslots);
                for (int sl = 0; sl < nslots; ++sl) {
                    if (n.tags->test(sl)) {
                        int& m = freshToNew[size_t(sl)];
                        if (m < 0) {
                            m = nNew++;
                            freshTouched.push_back(sl);
                        }
                        out[sl] = m;
                    } else {
                        const int raw = srcRegs[size_t(sl)];
                        if (raw < 0) {
                            out[sl] = -1;
                            continue;
                        }
                        int& m = oldToNew[size_t(raw)];
                        if (m < 0) {
                            m = nNew++;
                            oldTouched.push_back(raw);
                        }
                        out[sl] = m;
                    }
                }
                needMask |= need[size_t(n.pos)];
            }
        }
        if (nNew > kMaxTdfaRegs) throw RegexError("tdfa: register limit exceeded");
        maxRegs = std::max(maxRegs, nNew);
        const uint32_t prevCtx = nfa.behindBits(int(b)) & needMask;
        // the register program of the step: old registers in ascending order of their old names, then the fresh stamps in slot
        // order (the order std::map gave the general loop)
        regMovesBuf.clear();
        posDstsBuf.clear();
        std::sort(oldTouched.begin(), oldTouched.end());
        for (int r : oldTouched)
            if (oldToNew[size_t(r)] != r) regMovesBuf.emplace_back(oldToNew[size_t(r)], r);
        std::sort(freshTouched.begin(), freshTouched.end());
        for (int sl : freshTouched) posDstsBuf.push_back(freshToNew[size_t(sl)]);
        uint32_t listId = 0;
        if (!regMovesBuf.empty() || !posDstsBuf.empty()) {
            std::vector<uint16_t> sched = scheduleMoves(regMovesBuf, posDstsBuf);
            for (uint16_t w : sched)
                if ((w & 0xFF) == kRegTmp || (w >> 8) == kRegTmp) usedTmp = true;
            auto it = opListIds.find(sched);
            if (it == opListIds.end()) {
                if (opLists.size() >= 0xFFFF) throw RegexError("tdfa: too many distinct register programs");
                it = opListIds.emplace(sched, uint32_t(opLists.size())).first;
                opLists.push_back(sched);
            }
            listId = it->second;
        }
        // the target state: looked up by the key keyOf() would give it
        {
            const size_t perItem = sizeof(int) + size_t(nslots) * sizeof(int16_t) + 1;
            size_t linEntries = 0;
            segNames.clear();
            for (const Survivor& n : survivors)
                if (n.lin) linEntries += n.lin->size();
            keyBuf.resize(sizeof prevCtx + survivors.size() * perItem + linEntries * 3 * sizeof(int32_t));
            char* k = &keyBuf[0];
            std::memcpy(k, &prevCtx, sizeof prevCtx);
            k += sizeof prevCtx;
            for (size_t i = 0; i < survivors.size(); ++i) {
                std::memcpy(k, &survivors[i].pos, sizeof(int));
                k += sizeof(int);
                const int* regs = newRegs.data() + i * size_t(nslots);
                for (int sl = 0; sl < nslots; ++sl) {
                    const int16_t v = int16_t(regs[sl]);
                    std::memcpy(k, &v, sizeof v);
                    k += sizeof v;
                }
                if (survivors[i].lin)
                    for (const LinEntry& e : *survivors[i].lin) {  // segment ids are state-local names: canonical by first appearance
                        size_t name = 0;
                        while (name < segNames.size() && segNames[name] != e.seg) ++name;
                        if (name == segNames.size()) segNames.push_back(e.seg);
                        const int32_t v[3] = {e.g, int32_t(name), e.exited ? 1 : 0};
                        std::memcpy(k, v, sizeof v);
                        k += sizeof v;
                    }
                *k++ = '|';
            }
        }
        uint32_t tid;
        auto known = index.find(keyBuf);
        if (known != index.end()) {
            tid = known->second;
        } else {
            State Tn;
            Tn.items.resize(survivors.size());
            for (size_t i = 0; i < survivors.size(); ++i) {
                Tn.items[i].pos = survivors[i].pos;
                Tn.items[i].regs.assign(newRegs.begin() + long(i * size_t(nslots)), newRegs.begin() + long((i + 1) * size_t(nslots)));
                if (survivors[i].lin) {
                    Tn.items[i].lin = *survivors[i].lin;
                    for (LinEntry& e : Tn.items[i].lin) {
                        size_t name = 0;
                        while (segNames[name] != e.seg) ++name;
                        e.seg = int(name);
                    }
                }
            }
            Tn.nregs = nNew;
            Tn.prevCtx = prevCtx;
            tid = intern(std::move(Tn));
        }
        if (tid > 0xFFFF) throw RegexError("tdfa: state limit exceeded");
        return tid | (listId << 16);
    };
    // one (state, byte class) of a pattern without atomic groups -> the transition word (0: dead)
    auto stepFast = [&](uint32_t sid, int c) -> uint32_t {
        const unsigned b = classRep[size_t(c)];
        const uint32_t holds = states[sid].prevCtx | nfa.aheadBits(int(b));
        ++seenStamp;
        survivors.clear();
        {
            const State& S = states[sid];
            for (size_t k = 0; k < S.items.size(); ++k) {
                const size_t p = size_t(S.items[k].pos);
                const auto& lst = nfa.follow[p];
                pathWork += lst.size();
                const size_t v0 = viableStart[p * size_t(ncls) + size_t(c)], v1 = viableStart[p * size_t(ncls) + size_t(c) + 1];
                for (size_t v = v0; v < v1; ++v) {
                    const FollowPath& path = lst[viablePath[v]];
                    if (path.cond & ~holds) continue;
                    if (targetSeen[size_t(path.target)] == seenStamp) continue;
                    targetSeen[size_t(path.target)] = seenStamp;
                    survivors.push_back({path.target, int(k), &path.tags, nullptr});
                }
            }
        }
        if (pathWork > limits.maxPathWork)
            throw RegexError("tdfa: construction work limit (the automaton is too dense for a table; NFA engine)");
        if (survivors.empty()) return 0u;  // -> dead
        return emitTransition(sid, b);
    };
    // ... and of a pattern with atomic groups (the ordered commit)
    auto stepGeneral = [&](uint32_t sid, int c) -> uint32_t {
        const unsigned b = classRep[size_t(c)];
        std::vector<Cand> cands;
        uint32_t holds;
        {
            const State& S = states[sid];  // (not held across emitTransition: intern() may reallocate `states`)
            holds = S.prevCtx | nfa.aheadBits(int(b));
            const bool atomic = nfa.atomicCount > 0;
            ++seenStamp;
            for (size_t k = 0; k < S.items.size(); ++k) {
                pathWork += nfa.follow[S.items[k].pos].size();
                for (const auto& path : nfa.follow[S.items[k].pos]) {
                    const bool targetOk = path.target >= 0 && nfa.positions[path.target].has(b);
                    if (!atomic && (!targetOk || (path.cond & ~holds))) continue;  // cannot influence anything
                    if (!atomic) {  // without memberships the first path that reaches a position is the only one that counts
                        if (targetSeen[size_t(path.target)] == seenStamp) continue;
                        targetSeen[size_t(path.target)] = seenStamp;
                    }
                    // (a path that cannot consume the byte, crosses no group border or assertion and comes from an item without
                    // memberships closes nothing, opens nothing and is dropped at the end of commitAtomic: most paths of a large
                    // pattern, not worth a candidate)
                    if (!targetOk && path.atoms.empty() && S.items[k].lin.empty()) continue;
                    cands.push_back(Cand{path.target, int(k), path.tags, S.items[k].lin,
                                         atomic ? &path.atoms : nullptr, targetOk, path.cond});
                }
            }
        }
        if (pathWork > limits.maxPathWork)
            throw RegexError("tdfa: construction work limit (the automaton is too dense for a table; NFA engine)");
        std::vector<Cand> ni = commitAtomic(std::move(cands), holds);
        if (ni.empty()) return 0u;  // -> dead
        survivors.clear();
        for (const auto& n : ni) survivors.push_back({n.pos, n.src, &n.tags, n.lin.empty() ? nullptr : &n.lin});
        return emitTransition(sid, b);
    };
    auto step = [&](uint32_t sid, int c) -> uint32_t { return atomicPattern ? stepGeneral(sid, c) : stepFast(sid, c); };
    constexpr uint32_t kUnknown = 0xFFFFFFFFu;  // lazy form: a transition nobody has asked for yet
    uint64_t lazyComputed = 0, lazySteps = 0;
    bool lazyStopped = false;
    if (guide) {
        // the sample values, one after the other from the start state: only what they take is computed
        work.clear();
        auto rowOf = [&](uint32_t sid) -> std::vector<uint32_t>& {
            if (transRows.size() <= sid) transRows.resize(size_t(sid) + 1);
            if (transRows[sid].empty()) transRows[sid].assign(size_t(ncls), kUnknown);
            return transRows[sid];
        };
        transRows[0].assign(size_t(ncls), 0u);
        const size_t maxTableBytes = guide->maxTableBytes ? guide->maxTableBytes : (size_t(8) << 20);
        for (uint32_t v = 0; v < guide->n && !lazyStopped; ++v) {
            const uint8_t* p = guide->data + guide->off[v];
            const uint32_t L = guide->len[v];
            uint32_t st = T.startState;
            for (uint32_t i = 0; i < L && st != 0; ++i) {
                const int c = int(T.classMap[p[i]]);
                uint32_t w = rowOf(st)[size_t(c)];
                ++lazySteps;
                if (w == kUnknown) {
                    if ((states.size() + 2) * size_t(ncls) * 4 > maxTableBytes) {
                        lazyStopped = true;
                        break;
                    }
                    try {
                        w = step(st, c);
                    } catch (const RegexError&) {  // a limit: the table stays as far as it got
                        lazyStopped = true;
                        break;
                    }
                    rowOf(st)[size_t(c)] = w;
                    ++lazyComputed;
                }
                st = w & 0xFFFFu;
            }
        }
        // ... and every state a sample value has BEEN IN gets its whole row: a log line that takes the path of a sample value but
        // carries another byte class somewhere along it (a letter where the sample had digits, a rarer punctuation mark inside free
        // text) steps on a transition of a visited state -- which mostly leads back into visited states.  Without this a format's
        // table was still missing a fifth of its fresh values after 700 sample values of that format; the states this adds are not
        // walked further (their rows stay unknown until a value gets there).
        if (guide->completeRows && !lazyStopped) {
            const size_t visited = transRows.size();
            for (uint32_t sid = 1; sid < visited && !lazyStopped; ++sid) {
                if (transRows[sid].empty()) continue;
                for (int c = 0; c < ncls && !lazyStopped; ++c) {
                    if (transRows[sid][size_t(c)] != kUnknown) continue;
                    if ((states.size() + 2) * size_t(ncls) * 4 > maxTableBytes) {
                        lazyStopped = true;
                        break;
                    }
                    uint32_t w;
                    try {
                        w = step(sid, c);
                    } catch (const RegexError&) {
                        lazyStopped = true;
                        break;
                    }
                    transRows[sid][size_t(c)] = w;
                    ++lazyComputed;
                }
            }
        }
        work.clear();
    }
    while (!work.empty()) {
        const uint32_t sid = work.front();
        work.pop_front();
        std::vector<uint32_t> row(size_t(ncls), 0u);
        for (int c = 0; c < ncls; ++c) row[size_t(c)] = step(sid, c);
        if (transRows.size() <= sid) transRows.resize(sid + 1);
        transRows[sid] = std::move(row);
    }
    const uint32_t nReal = uint32_t(states.size());
    T.nStates = nReal + (guide ? 1u : 0u);  // lazy form: one more state, the MISS sink
    if (T.nStates > 0xFFFFu) throw RegexError("tdfa: state limit exceeded");
    T.trans.resize(size_t(T.nStates) * size_t(ncls));
    uint64_t lazyUnknown = 0;
    for (uint32_t s = 0; s < nReal; ++s)
        for (int c = 0; c < ncls; ++c) {
            uint32_t w = 0;
            if (!guide) w = transRows[s][size_t(c)];
            else if (s == 0) w = 0;
            else if (s < transRows.size() && !transRows[s].empty()) w = transRows[s][size_t(c)];
            else w = kUnknown;
            if (w == kUnknown) {
                w = nReal;  // -> MISS, no register program
                ++lazyUnknown;
            }
            T.trans[size_t(s) * size_t(ncls) + size_t(c)] = w;
        }
    if (guide) {
        for (int c = 0; c < ncls; ++c) T.trans[size_t(nReal) * size_t(ncls) + size_t(c)] = nReal;  // a sink
        T.missState = nReal;
    }

    const int tmpReg = maxRegs;
    T.nRegs = uint32_t(maxRegs + (usedTmp ? 1 : 0));
    if (T.nRegs == 0) T.nRegs = 1;
    T.opsStart.push_back(0);
    for (const auto& lst : opLists) {
        if (!lst.empty()) {
            T.ops.push_back(uint16_t(lst.size()));
            for (uint16_t w : lst) {
                int d = w & 0xFF, s = w >> 8;
                if (d == kRegTmp) d = tmpReg;
                if (s == kRegTmp) s = tmpReg;
                T.ops.push_back(uint16_t(d | (s << 8)));
            }
        }
        T.opsStart.push_back(uint32_t(T.ops.size()));
    }
    // opsStart[id] = start of list id (list 0 is empty so opsStart[0] == opsStart[1] == 0)

    // ---- acceptance at end of input: first item (priority order) with a MATCH path whose assertions hold at END
    T.finalId.assign(T.nStates, 0xFFFF);
    std::map<std::vector<uint8_t>, uint16_t> finIds;
    for (uint32_t s = 1; s < nReal; ++s) {
        const State& S = states[s];
        const uint32_t holds = S.prevCtx | nfa.aheadBits(kEdge);
        std::vector<Cand> cands;
        const bool atomic = nfa.atomicCount > 0;
        for (size_t k = 0; k < S.items.size(); ++k)
            for (const auto& path : nfa.follow[size_t(S.items[k].pos)]) {
                const bool targetOk = path.target == kMatchTarget;
                if (!atomic && (!targetOk || (path.cond & ~holds))) continue;
                // pos = a unique id: survivors must not collapse here, the first one wins
                cands.push_back(Cand{int(cands.size()), int(k), path.tags, S.items[k].lin,
                                     atomic ? &path.atoms : nullptr, targetOk, path.cond});
            }
        const std::vector<Cand> winners = commitAtomic(std::move(cands), holds);
        if (!winners.empty()) {
            const Cand& wn = winners.front();  // highest-priority thread that is in MATCH when the input is exhausted
            std::vector<uint8_t> fm(size_t(nslots) ? size_t(nslots) : 1, kRegNone);
            for (int sl = 0; sl < nslots; ++sl) {
                if (wn.tags.test(sl)) fm[size_t(sl)] = kRegPos;
                else if (S.items[size_t(wn.src)].regs[size_t(sl)] >= 0)
                    fm[size_t(sl)] = uint8_t(S.items[size_t(wn.src)].regs[size_t(sl)]);
            }
            auto it = finIds.find(fm);
            if (it == finIds.end()) {
                it = finIds.emplace(fm, uint16_t(finIds.size())).first;
                T.finalMap.insert(T.finalMap.end(), fm.begin(), fm.end());
            }
            T.finalId[s] = it->second;
        }
    }
    if (T.finalMap.empty()) T.finalMap.assign(size_t(nslots) ? size_t(nslots) : 1, kRegNone);
    if (guide) {
        // the MISS state gets a final row of its OWN NUMBER (content: nothing captured): the minimisation tells states apart by that
        // number first, so the sink is never merged with the dead state or an absorbing accept; the kernels test for the MISS state
        // before they look at final rows
        T.finalId[T.missState] = uint16_t(T.finalMap.size() / (size_t(nslots) ? size_t(nslots) : 1));
        T.finalMap.insert(T.finalMap.end(), size_t(nslots) ? size_t(nslots) : 1, kRegNone);
    }
    eliminateDeadStores(T);
    minimizeTdfaStates(T);
    if (report) {
        report->statesBuilt = nReal;
        report->statesKept = T.nStates;
        report->transitionsComputed = lazyComputed;
        report->transitionsUnknown = lazyUnknown;
        report->stepsWalked = lazySteps;
        report->stopped = lazyStopped;
    }
    return T;
}

}  // namespace lcregex
