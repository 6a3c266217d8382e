// atomic_elide.cpp -- atomic groups that cannot change a result are turned into plain groups before the automata are built.
//
// (?>X) keeps the first way X matches and never retries another one.  That costs both engines dearly: the NFA kernel runs an ORDERED
// commit pass on one lane for every byte a thread spends inside an unsettled group (nfa_kernel.hpp nfaAtomicStep: a 4 KiB quoted
// string behind Grok's QUOTEDSTRING took 250 ms), and the determinisation has to track which alternatives a commit closes.  Most
// atomic groups of real pattern libraries are performance hints for backtrackers, not semantics -- QUOTEDSTRING, YEAR "(?>\d\d){1,2}"
// (in every time stamp), WINPATH's "(?>[A-Za-z]+:|\\)".
//
// The rule (exact, not a heuristic).  Let G = (?>X) contain no capturing group and no look-around (a look-BEHIND that leads X is
// peeled off first: it looks at the byte before the group and is the same test in both forms), and let S = X with every atomic group
// inside it made plain.  If
//     (1)  L(G) = L(S)       -- the strings G matches exactly (inner atomic groups honoured) are the strings S matches exactly, and
//     (2)  L(S) is prefix-free -- no match of S is a proper prefix of another match of S,
// then replacing G by (?:S) changes no match and no capture of any pattern G occurs in:  entered at offset s, G commits to the end e of
// its first way and text[s:e) is in L(G) (ways of higher priority failed on the text and fail the same way on the truncated text: there
// is no look-ahead to tell the difference); every way of S ends at some e' with text[s:e') in L(S) = L(G); two different ends would be
// a proper prefix pair, so all ways of S and the one way of G end at the same offset, and exist together or not at all.  What follows
// the group is tried from that one offset in both forms; retrying another way of S behind a failure repeats the same failure.  Nothing
// inside the group is captured, so which way was taken is unobservable.
// Both languages are decided on the tagged DFAs the TDFA builder makes for G and S as whole-line patterns (tdfa.cpp honours atomic
// groups: commitAtomic); a group whose automaton exceeds small limits is simply kept.  Groups that stay (BASE10NUM: "12" is a prefix
// of "12.5") are handled by the engines as before.  The oracle (oracle/bt_regex.c) implements atomic groups natively and knows nothing
// of this pass: tests/test_gpu_parity.py::test_atomic_groups_* and the Grok suites compare against it.
#include <deque>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "follow_nfa.hpp"
#include "regex_ast.hpp"
#include "tdfa.hpp"

namespace lcregex {
namespace {

std::unique_ptr<Node> cloneNode(const Node& n, bool stripAtomics) {
    auto c = std::make_unique<Node>();
    c->kind = (stripAtomics && n.kind == Node::Atomic) ? Node::Group : n.kind;
    c->set = n.set;
    c->min = n.min;
    c->max = n.max;
    c->greedy = n.greedy;
    c->capture = n.capture;
    c->look = n.look;
    c->aheadSeq = n.aheadSeq;
    c->aheadNegative = n.aheadNegative;
    c->window = n.window;
    c->runCapture = n.runCapture;
    for (const auto& k : n.kids) c->kids.push_back(cloneNode(*k, stripAtomics));
    return c;
}

void serialize(const Node& n, std::string& out) {
    out.push_back(char('A' + int(n.kind)));
    switch (n.kind) {
        case Node::Set:
            for (uint64_t w : n.set.w) out.append(reinterpret_cast<const char*>(&w), 8);
            break;
        case Node::Repeat:
            out += std::to_string(n.min) + "," + std::to_string(n.max) + (n.greedy ? "g" : "l");
            break;
        case Node::Group: out += std::to_string(n.capture) + (n.runCapture ? "r" : ""); break;
        case Node::Assert:
            out.push_back(n.look.behind ? 'b' : 'a');
            out.push_back(n.look.edgeOk ? '1' : '0');
            for (uint64_t w : n.look.set.w) out.append(reinterpret_cast<const char*>(&w), 8);
            break;
        default: break;
    }
    out.push_back('(');
    for (const auto& k : n.kids) serialize(*k, out);
    out.push_back(')');
}

bool hasAtomic(const Node& n) {
    if (n.kind == Node::Atomic) return true;
    for (const auto& k : n.kids)
        if (hasAtomic(*k)) return true;
    return false;
}

// nothing whose effect could be seen from outside the group or that looks outside it
bool plainBody(const Node& n) {
    if (n.kind == Node::Assert) return false;
    if (n.kind == Node::Group && (n.capture != 0 || n.runCapture)) return false;
    if (!n.aheadSeq.empty()) return false;
    for (const auto& k : n.kids)
        if (!plainBody(*k)) return false;
    return true;
}

struct Dfa {
    uint32_t nStates = 0, nClasses = 0, start = 0;
    std::vector<uint8_t> classMap;
    std::vector<uint32_t> next;   // [nStates][nClasses]
    std::vector<uint8_t> accept;  // [nStates]
};

bool buildWholeLineDfa(std::unique_ptr<Node> root, Dfa& out) {
    try {
        ParsedRegex re;
        re.root = std::move(root);
        re.groupCount = 0;
        re.groupNames.assign(1, std::string());
        const FollowNfa nfa = buildFollowNfa(re);
        TdfaLimits lim;
        lim.maxStates = 1024;
        lim.maxPathWork = 1u << 20;
        lim.maxCommitWork = 2u << 20;
        lim.ldsWindow = false;
        const TdfaTables t = buildTdfa(nfa, lim);
        out.nStates = t.nStates;
        out.nClasses = t.nClasses;
        out.start = t.startState;
        out.classMap = t.classMap;
        out.next.resize(size_t(t.nStates) * t.nClasses);
        for (size_t i = 0; i < out.next.size(); ++i) out.next[i] = t.trans[i] & 0xFFFFu;
        out.accept.resize(t.nStates);
        for (uint32_t s = 0; s < t.nStates; ++s) out.accept[s] = s != 0 && t.finalId[s] != 0xFFFF;
        return true;
    } catch (const RegexError&) {
        return false;
    }
}

bool sameLanguage(const Dfa& a, const Dfa& b) {
    // the byte classes both automata tell apart
    std::map<std::pair<uint8_t, uint8_t>, unsigned> joint;
    for (unsigned c = 0; c < 256; ++c) joint.emplace(std::make_pair(a.classMap[c], b.classMap[c]), c);
    std::map<std::pair<uint32_t, uint32_t>, bool> seen;
    std::deque<std::pair<uint32_t, uint32_t>> todo{{a.start, b.start}};
    seen[{a.start, b.start}] = true;
    while (!todo.empty()) {
        const auto [sa, sb] = todo.front();
        todo.pop_front();
        if (a.accept[sa] != b.accept[sb]) return false;
        if (sa == 0 && sb == 0) continue;
        for (const auto& kv : joint) {
            const uint32_t na = sa ? a.next[size_t(sa) * a.nClasses + kv.first.first] : 0;
            const uint32_t nb = sb ? b.next[size_t(sb) * b.nClasses + kv.first.second] : 0;
            if (seen.emplace(std::make_pair(na, nb), true).second) todo.emplace_back(na, nb);
            if (seen.size() > 200000) return false;  // (never: both automata are small)
        }
    }
    return true;
}

// no accepting state is reachable from an accepting state by one or more bytes
bool prefixFree(const Dfa& d) {
    // states from which an accepting state is reachable in >= 0 steps
    std::vector<uint8_t> canAccept(d.nStates, 0);
    std::vector<std::vector<uint32_t>> pred(d.nStates);
    for (uint32_t s = 1; s < d.nStates; ++s)
        for (uint32_t c = 0; c < d.nClasses; ++c) pred[d.next[size_t(s) * d.nClasses + c]].push_back(s);
    std::deque<uint32_t> todo;
    for (uint32_t s = 1; s < d.nStates; ++s)
        if (d.accept[s]) {
            canAccept[s] = 1;
            todo.push_back(s);
        }
    while (!todo.empty()) {
        const uint32_t s = todo.front();
        todo.pop_front();
        for (uint32_t p : pred[s])
            if (!canAccept[p]) {
                canAccept[p] = 1;
                todo.push_back(p);
            }
    }
    // (only states reachable from the start matter; unreachable ones do not exist after the builder's minimisation)
    for (uint32_t s = 1; s < d.nStates; ++s) {
        if (!d.accept[s]) continue;
        for (uint32_t c = 0; c < d.nClasses; ++c) {
            const uint32_t n = d.next[size_t(s) * d.nClasses + c];
            if (n != 0 && canAccept[n]) return false;
        }
    }
    return true;
}

struct Elider {
    std::map<std::string, bool> memo;
    int elided = 0;

    bool redundant(const Node& body) {
        std::string key;
        serialize(body, key);
        auto it = memo.find(key);
        if (it != memo.end()) return it->second;
        bool ok = false;
        Dfa strict, plain;
        auto wrapped = std::make_unique<Node>();
        wrapped->kind = Node::Atomic;
        wrapped->kids.push_back(cloneNode(body, false));
        if (buildWholeLineDfa(std::move(wrapped), strict) && buildWholeLineDfa(cloneNode(body, true), plain))
            ok = prefixFree(plain) && sameLanguage(strict, plain);
        memo.emplace(std::move(key), ok);
        return ok;
    }

    void visit(Node& n) {
        if (n.kind == Node::Atomic && !n.kids.empty()) {
            Node& x = *n.kids[0];
            // look-behinds that lead the body look at the byte before the group: the same test in both forms
            size_t lead = 0;
            if (x.kind == Node::Cat)
                while (lead < x.kids.size() && x.kids[lead]->kind == Node::Assert && x.kids[lead]->look.behind) ++lead;
            bool ok;
            if (lead == 0) {
                ok = plainBody(x) && redundant(x);
            } else {
                auto rest = std::make_unique<Node>();
                rest->kind = Node::Cat;
                for (size_t i = lead; i < x.kids.size(); ++i) rest->kids.push_back(cloneNode(*x.kids[i], false));
                ok = plainBody(*rest) && redundant(*rest);
            }
            if (ok) {
                n.kids[0] = cloneNode(x, true);
                n.kind = Node::Group;
                n.capture = 0;
                ++elided;
                return;  // (everything inside went plain with it)
            }
        }
        for (auto& k : n.kids) visit(*k);
    }
};

}  // namespace

int elideRedundantAtomics(ParsedRegex& re) {
    if (!re.root || !hasAtomic(*re.root)) return 0;
    Elider e;
    e.visit(*re.root);
    return e.elided;
}

}  // namespace lcregex
