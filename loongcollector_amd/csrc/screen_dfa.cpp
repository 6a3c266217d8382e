// screen_dfa.cpp -- plain (untagged, unordered) DFA for SCREENS: "can this line contain a match at all?".
//
// A screen only answers yes or no, so the determinisation need not keep what the tagged construction (tdfa.cpp) keeps:
// thread order, register maps, prev-byte context.  What it CAN do instead is forget: in a Grok pattern relaxed for
// screening (regex_handle.cpp lcCompileRelaxedScreen) the fields between the literals are "anything" -- universal loops
// (?s:.)* -- and once such a loop is alive, every position that can only reach MATCH THROUGH that loop says nothing new: the
// loop eats whatever that position would have eaten and goes on from there.  Dropping those positions (the loop
// post-dominates them in the follow graph) turns "X.*Y.*Z" from a product of the progress through X, Y and Z into their sum;
// without it the subset construction of such a pattern under a search wrapper passes 10^5 states.
//
// Output: TdfaTables without register programs, consumed by packTdfaBlob / the TDFA kernels with ngroups = 0.
#include <algorithm>
#include <map>

#include "tdfa.hpp"

namespace lcregex {

TdfaTables buildScreenDfaUncached(const FollowNfa& nfa, const TdfaLimits& limits) {
    if (nfa.condsUsed != 0 || nfa.atomicCount > 0) throw RegexError("screen dfa: assertions / atomic groups are not relaxed away");
    const int npos = int(nfa.positions.size());
    const int startNode = npos, exitNode = npos + 1, nNodes = npos + 2;
    TdfaTables T;
    T.nSlots = uint32_t(nfa.slotCount());

    // ---- byte classes
    T.classMap.assign(256, 0);
    std::vector<unsigned> classRep;
    {
        std::map<std::vector<bool>, int> sig2cls;
        for (unsigned b = 0; b < 256; ++b) {
            std::vector<bool> sig;
            sig.reserve(size_t(npos));
            for (int p = 0; p < npos; ++p) sig.push_back(nfa.positions[size_t(p)].has(b));
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
    const uint32_t maxStates =
        limits.ldsWindow ? std::min<uint32_t>(limits.maxStates, (65536u - 320u) / (uint32_t(ncls + 1) * 4u)) : limits.maxStates;

    // ---- follow graph: successors per node, MATCH = exitNode
    std::vector<std::vector<int>> succ(static_cast<size_t>(nNodes));
    std::vector<char> accepts(size_t(npos) + 1, 0);  // node has a MATCH path
    for (int p = 0; p <= npos; ++p) {
        for (const auto& path : nfa.follow[size_t(p)]) {
            const int t = path.target == kMatchTarget ? exitNode : path.target;
            if (t == exitNode) accepts[size_t(p)] = 1;
            if (std::find(succ[size_t(p)].begin(), succ[size_t(p)].end(), t) == succ[size_t(p)].end()) succ[size_t(p)].push_back(t);
        }
    }
    // universal loops: take every byte, loop on themselves
    std::vector<int> universal;
    for (int p = 0; p < npos; ++p)
        if (nfa.positions[size_t(p)] == ByteSet::all() && std::find(succ[size_t(p)].begin(), succ[size_t(p)].end(), p) != succ[size_t(p)].end())
            universal.push_back(p);
    // post-dominators, only with respect to the universal loops: dominatedBy[u] = nodes every path from which to MATCH
    // passes through u.  Fixpoint of  reachesExitAvoiding(u): a node is NOT dominated iff it reaches exit without u.
    const size_t words = (size_t(nNodes) + 63) / 64;
    std::vector<std::vector<uint64_t>> dominated(universal.size(), std::vector<uint64_t>(words, 0));
    {
        std::vector<std::vector<int>> pred(static_cast<size_t>(nNodes));
        for (int p = 0; p < nNodes; ++p)
            for (int t : succ[size_t(p)]) pred[size_t(t)].push_back(p);
        for (size_t ui = 0; ui < universal.size(); ++ui) {
            const int u = universal[ui];
            std::vector<char> escapes(static_cast<size_t>(nNodes), 0);  // reaches MATCH without passing through u
            std::vector<int> work{exitNode};
            escapes[size_t(exitNode)] = 1;
            while (!work.empty()) {
                const int x = work.back();
                work.pop_back();
                for (int q : pred[size_t(x)])
                    if (q != u && !escapes[size_t(q)]) {
                        escapes[size_t(q)] = 1;
                        work.push_back(q);
                    }
            }
            for (int q = 0; q <= npos; ++q)
                if (q != u && !escapes[size_t(q)]) dominated[ui][size_t(q) >> 6] |= uint64_t(1) << (q & 63);
        }
    }

    // ---- subset construction over sorted position sets
    typedef std::vector<int> Set;
    std::map<Set, uint32_t> ids;
    std::vector<Set> states;
    states.emplace_back();  // 0 = dead
    auto prune = [&](Set s) {
        std::sort(s.begin(), s.end());
        s.erase(std::unique(s.begin(), s.end()), s.end());
        // acceptance is certain once the search wrapper's suffix loop is alive
        if (nfa.searchSuffix >= 0 && std::binary_search(s.begin(), s.end(), nfa.searchSuffix)) return Set{nfa.searchSuffix};
        for (size_t ui = 0; ui < universal.size(); ++ui) {
            if (!std::binary_search(s.begin(), s.end(), universal[ui])) continue;
            const auto& dom = dominated[ui];
            s.erase(std::remove_if(s.begin(), s.end(), [&](int q) { return (dom[size_t(q) >> 6] >> (q & 63)) & 1; }), s.end());
        }
        return s;
    };
    auto intern = [&](Set s) -> uint32_t {
        if (s.empty()) return 0;
        auto it = ids.find(s);
        if (it != ids.end()) return it->second;
        if (states.size() >= maxStates) throw RegexError("screen dfa: state limit exceeded");
        const uint32_t id = uint32_t(states.size());
        ids.emplace(s, id);
        states.push_back(std::move(s));
        return id;
    };
    T.startState = intern(Set{startNode});
    if (nfa.searchPrefix >= 0) {
        const uint32_t resumed = intern(prune(Set{nfa.searchPrefix}));
        T.startAfter.assign(size_t(ncls), resumed);
    }
    std::vector<std::vector<uint32_t>> rows;
    rows.emplace_back(size_t(ncls), 0u);
    for (uint32_t sid = 1; sid < states.size(); ++sid) {
        const Set cur = states[sid];  // (a copy: intern() grows `states`)
        std::vector<uint32_t> row(size_t(ncls), 0u);
        for (int c = 0; c < ncls; ++c) {
            Set next;
            for (int p : cur)
                for (int t : succ[size_t(p)])
                    if (t != exitNode && nfa.positions[size_t(t)].has(classRep[size_t(c)])) next.push_back(t);
            row[size_t(c)] = intern(prune(std::move(next)));
            if (row[size_t(c)] > 0xFFFF) throw RegexError("screen dfa: state limit exceeded");
        }
        if (rows.size() <= sid) rows.resize(sid + 1);
        rows[sid] = std::move(row);
    }
    T.nStates = uint32_t(states.size());
    T.trans.resize(size_t(T.nStates) * size_t(ncls));
    for (uint32_t s = 0; s < T.nStates; ++s)
        for (int c = 0; c < ncls; ++c) T.trans[size_t(s) * size_t(ncls) + size_t(c)] = rows[s][size_t(c)];
    T.nRegs = 1;
    T.opsStart = {0, 0};  // only the empty list
    T.finalId.assign(T.nStates, 0xFFFF);
    T.finalMap.assign(T.nSlots ? T.nSlots : 1, kRegNone);
    for (uint32_t s = 1; s < T.nStates; ++s)
        for (int p : states[s])
            if (accepts[size_t(p)]) T.finalId[s] = 0;
    minimizeTdfaStates(T);
    return T;
}

}  // namespace lcregex
