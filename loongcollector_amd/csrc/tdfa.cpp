// tdfa.cpp -- follow NFA -> tagged DFA tables.  See tdfa.hpp.
#include "tdfa.hpp"

#include <algorithm>
#include <deque>
#include <map>
#include <string>
#include <unordered_map>

namespace lcregex {

namespace {

constexpr uint8_t kRegTmp = 0xFD;  // placeholder, patched to the real scratch register at the end
constexpr int kFreshBase = 1 << 20;

struct Item {
    int pos;
    std::vector<int> regs;  // per slot: register id, or -1
};
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

TdfaTables buildTdfa(const FollowNfa& nfa, const TdfaLimits& limits) {
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

    std::vector<State> states;
    std::unordered_map<std::string, uint32_t> index;
    std::deque<uint32_t> work;
    states.emplace_back();  // state 0 = dead
    auto intern = [&](State&& s) -> uint32_t {
        std::string k = keyOf(s);
        auto it = index.find(k);
        if (it != index.end()) return it->second;
        if (states.size() >= limits.maxStates) throw RegexError("tdfa: state limit exceeded");
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

    std::map<std::vector<uint16_t>, uint32_t> opListIds;
    std::vector<std::vector<uint16_t>> opLists(1);  // id 0 = empty
    std::vector<std::vector<uint32_t>> transRows;   // filled per state id
    transRows.emplace_back(size_t(ncls), 0u);       // dead row
    int maxRegs = 0;
    bool usedTmp = false;

    struct NewItem {
        int pos, src;
        uint64_t tags;
    };
    std::vector<char> seen;

    while (!work.empty()) {
        uint32_t sid = work.front();
        work.pop_front();
        if (transRows.size() <= sid) transRows.resize(sid + 1);
        std::vector<uint32_t> row(size_t(ncls), 0u);
        for (int c = 0; c < ncls; ++c) {
            const State& S = states[sid];  // re-fetched each class: intern() may reallocate `states`
            const unsigned b = classRep[c];
            const uint32_t holds = S.prevCtx | nfa.aheadBits(int(b));
            std::vector<NewItem> ni;
            seen.assign(size_t(npos), 0);
            for (size_t k = 0; k < S.items.size(); ++k) {
                for (const auto& path : nfa.follow[S.items[k].pos]) {
                    if (path.target < 0 || seen[path.target]) continue;
                    if (!nfa.positions[path.target].has(b)) continue;
                    if (path.cond & ~holds) continue;
                    seen[path.target] = 1;
                    ni.push_back({path.target, int(k), path.tags});
                }
            }
            if (ni.empty()) continue;  // -> dead
            State Tn;
            std::map<int, int> rename;
            uint32_t needMask = 0;
            for (const auto& n : ni) {
                Item it;
                it.pos = n.pos;
                it.regs.resize(size_t(nslots));
                for (int s = 0; s < nslots; ++s) {
                    int raw = ((n.tags >> s) & 1) ? kFreshBase + s : S.items[size_t(n.src)].regs[size_t(s)];
                    if (raw < 0) {
                        it.regs[size_t(s)] = -1;
                        continue;
                    }
                    auto r = rename.find(raw);
                    if (r == rename.end()) r = rename.emplace(raw, int(rename.size())).first;
                    it.regs[size_t(s)] = r->second;
                }
                needMask |= need[size_t(n.pos)];
                Tn.items.push_back(std::move(it));
            }
            Tn.nregs = int(rename.size());
            if (Tn.nregs > kMaxTdfaRegs) throw RegexError("tdfa: register limit exceeded");
            maxRegs = std::max(maxRegs, Tn.nregs);
            Tn.prevCtx = nfa.behindBits(int(b)) & needMask;

            std::vector<std::pair<int, int>> regMoves;
            std::vector<int> posDsts;
            for (const auto& kv : rename) {
                if (kv.first >= kFreshBase) posDsts.push_back(kv.second);
                else if (kv.first != kv.second) regMoves.emplace_back(kv.second, kv.first);
            }
            std::vector<uint16_t> sched = scheduleMoves(std::move(regMoves), posDsts);
            uint32_t listId = 0;
            if (!sched.empty()) {
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
            uint32_t tid = intern(std::move(Tn));
            if (tid > 0xFFFF) throw RegexError("tdfa: state limit exceeded");
            row[size_t(c)] = tid | (listId << 16);
        }
        if (transRows.size() <= sid) transRows.resize(sid + 1);
        transRows[sid] = std::move(row);
    }

    T.nStates = uint32_t(states.size());
    T.trans.resize(size_t(T.nStates) * size_t(ncls));
    for (uint32_t s = 0; s < T.nStates; ++s)
        for (int c = 0; c < ncls; ++c) T.trans[size_t(s) * size_t(ncls) + size_t(c)] = transRows[s][size_t(c)];

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
    for (uint32_t s = 1; s < T.nStates; ++s) {
        const State& S = states[s];
        const uint32_t holds = S.prevCtx | nfa.aheadBits(kEdge);
        bool done = false;
        for (size_t k = 0; k < S.items.size() && !done; ++k) {
            for (const auto& path : nfa.follow[size_t(S.items[k].pos)]) {
                if (path.target != kMatchTarget) continue;
                if (path.cond & ~holds) continue;
                std::vector<uint8_t> fm(size_t(nslots) ? size_t(nslots) : 1, kRegNone);
                for (int sl = 0; sl < nslots; ++sl) {
                    if ((path.tags >> sl) & 1) fm[size_t(sl)] = kRegPos;
                    else if (S.items[k].regs[size_t(sl)] >= 0) fm[size_t(sl)] = uint8_t(S.items[k].regs[size_t(sl)]);
                }
                auto it = finIds.find(fm);
                if (it == finIds.end()) {
                    it = finIds.emplace(fm, uint16_t(finIds.size())).first;
                    T.finalMap.insert(T.finalMap.end(), fm.begin(), fm.end());
                }
                T.finalId[s] = it->second;
                done = true;
                break;
            }
        }
    }
    if (T.finalMap.empty()) T.finalMap.assign(size_t(nslots) ? size_t(nslots) : 1, kRegNone);
    return T;
}

}  // namespace lcregex
