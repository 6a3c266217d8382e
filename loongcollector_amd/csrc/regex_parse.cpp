// regex_parse.cpp -- recursive-descent parser: pattern bytes -> lcregex::Node tree.
// See regex_ast.hpp for what this replaces on the reference path.
#include <cstdlib>
#include <cstring>

#include "regex_ast.hpp"

namespace lcregex {

bool isWordByte(unsigned c) {
    return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_';
}
bool isLineSeparator(unsigned c) { return c == '\n' || c == '\r' || c == '\f'; }

namespace {

using NodePtr = std::unique_ptr<Node>;

NodePtr mk(Node::Kind k) {
    auto n = std::make_unique<Node>();
    n->kind = k;
    return n;
}

void foldCase(ByteSet& s) {
    for (unsigned c = 'a'; c <= 'z'; ++c) {
        if (s.has(c) || s.has(c - 32)) {
            s.add(c);
            s.add(c - 32);
        }
    }
}

// shorthand classes keyed by the lower-case escape letter
bool shorthand(char letter, ByteSet& out) {
    switch (letter) {
        case 'd': out.addRange('0', '9'); return true;
        case 'w':
            out.addRange('0', '9');
            out.addRange('a', 'z');
            out.addRange('A', 'Z');
            out.add('_');
            return true;
        case 's':
            out.add(' ');
            out.addRange(9, 13);
            return true;
        case 'h':
            out.add(' ');
            out.add('\t');
            return true;
        case 'v': out.addRange(10, 13); return true;
        case 'l': out.addRange('a', 'z'); return true;
        case 'u': out.addRange('A', 'Z'); return true;
        default: return false;
    }
}

bool posixClass(std::string_view name, ByteSet& out) {
    struct Entry {
        const char* name;
        void (*fill)(ByteSet&);
    };
    static const Entry table[] = {
        {"alpha", [](ByteSet& s) { s.addRange('a', 'z'); s.addRange('A', 'Z'); }},
        {"digit", [](ByteSet& s) { s.addRange('0', '9'); }},
        {"d", [](ByteSet& s) { s.addRange('0', '9'); }},
        {"alnum", [](ByteSet& s) { s.addRange('a', 'z'); s.addRange('A', 'Z'); s.addRange('0', '9'); }},
        {"upper", [](ByteSet& s) { s.addRange('A', 'Z'); }},
        {"u", [](ByteSet& s) { s.addRange('A', 'Z'); }},
        {"lower", [](ByteSet& s) { s.addRange('a', 'z'); }},
        {"l", [](ByteSet& s) { s.addRange('a', 'z'); }},
        {"space", [](ByteSet& s) { s.add(' '); s.addRange(9, 13); }},
        {"s", [](ByteSet& s) { s.add(' '); s.addRange(9, 13); }},
        {"blank", [](ByteSet& s) { s.add(' '); s.add('\t'); }},
        {"punct", [](ByteSet& s) { s.addRange(33, 47); s.addRange(58, 64); s.addRange(91, 96); s.addRange(123, 126); }},
        {"print", [](ByteSet& s) { s.addRange(32, 126); }},
        {"graph", [](ByteSet& s) { s.addRange(33, 126); }},
        {"cntrl", [](ByteSet& s) { s.addRange(0, 31); s.add(127); }},
        {"xdigit", [](ByteSet& s) { s.addRange('0', '9'); s.addRange('a', 'f'); s.addRange('A', 'F'); }},
        {"word", [](ByteSet& s) { shorthand('w', s); }},
        {"w", [](ByteSet& s) { shorthand('w', s); }},
    };
    for (const auto& e : table) {
        if (name == e.name) {
            e.fill(out);
            return true;
        }
    }
    return false;
}

int hexDigit(unsigned c) {
    if (c >= '0' && c <= '9') return int(c - '0');
    if (c >= 'a' && c <= 'f') return int(c - 'a' + 10);
    if (c >= 'A' && c <= 'F') return int(c - 'A' + 10);
    return -1;
}

class Parser {
public:
    Parser(std::string_view p, Syntax s) : mPat(p), mSyn(s) { mNames.emplace_back(); }

    ParsedRegex run() {
        NodePtr root = alternation(0);
        if (!atEnd()) bail(peek() == ')' ? "unmatched )" : "unexpected character");
        ParsedRegex out;
        out.root = std::move(root);
        out.groupCount = mGroups;
        out.groupNames = std::move(mNames);
        for (int r : mBackRefs) {  // (boost: error_backref at compile time)
            if (r > mGroups) throw RegexError("invalid back reference: \\" + std::to_string(r) + " with " + std::to_string(mGroups) + " groups");
        }
        for (const std::string& w : mNamedRefs) {  // (same-named groups are ONE group to regexp2 and boost: which text would come back?)
            int count = 0;
            for (const std::string& nm : out.groupNames) count += nm == w;
            if (count != 1) throw RegexError("back-reference to a name that several groups carry: unsupported");
        }
        out.hasBackRef = !mBackRefs.empty();
        out.hasGeneralLook = mGeneralLook;
        return out;
    }

private:
    std::string_view mPat;
    size_t mPos = 0;
    Syntax mSyn;
    int mGroups = 0;
    std::vector<std::string> mNames;
    std::vector<int> mBackRefs;
    std::vector<std::string> mNamedRefs;
    bool mGeneralLook = false;

    // every match of `n` has this length, or -1 (a look-behind body must have one: Perl, PCRE, regexp2 and boost agree)
    static int fixedLen(const Node& n) {
        switch (n.kind) {
            case Node::Empty:
            case Node::Assert:
            case Node::Look: return 0;
            case Node::Set: return 1;
            case Node::Cat: {
                int sum = 0;
                for (const auto& k : n.kids) {
                    const int a = fixedLen(*k);
                    if (a < 0) return -1;
                    sum += a;
                }
                return sum;
            }
            case Node::Alt: {
                int len = -2;
                for (const auto& k : n.kids) {
                    const int a = fixedLen(*k);
                    if (a < 0 || (len != -2 && a != len)) return -1;
                    len = a;
                }
                return len < 0 ? 0 : len;
            }
            case Node::Repeat: {
                const int a = fixedLen(*n.kids[0]);
                return (a < 0 || n.min != n.max) ? -1 : a * n.min;
            }
            case Node::Group: return n.runCapture ? -1 : fixedLen(*n.kids[0]);
            case Node::Atomic: return fixedLen(*n.kids[0]);
            case Node::Cond: {
                const int a = fixedLen(*n.kids[0]), b = fixedLen(*n.kids[1]);
                return (a < 0 || a != b) ? -1 : a;
            }
            case Node::BackRef: return -1;
        }
        return -1;
    }
    // the general form (Node::Look): runs on the device backtracking engine.  Not under Grok's dialect: its matcher plans automata.
    NodePtr generalLook(NodePtr body, bool behind, bool negative, const char* why) {
        if (mSyn.regexp2 && !mSyn.namedOnly) bail(why);  // (the Go regex plugin: RE2 has no look-arounds beyond what the parser decides)
        int k = 0;
        if (behind) {
            k = fixedLen(*body);
            if (k < 0) bail("unsupported: look-behind body of variable length");  // (boost refuses it too)
        }
        auto n = mk(Node::Look);
        n->look.behind = behind;
        n->aheadNegative = negative;
        n->min = k;
        n->kids.push_back(std::move(body));
        mGeneralLook = true;
        return n;
    }

    bool atEnd() const { return mPos >= mPat.size(); }
    unsigned peek(size_t ahead = 0) const { return static_cast<unsigned char>(mPat[mPos + ahead]); }
    bool has(size_t ahead) const { return mPos + ahead < mPat.size(); }
    [[noreturn]] void bail(const std::string& what) const {
        throw RegexError(what + " at offset " + std::to_string(mPos));
    }

    void skipFreeSpacing() {
        if (!mSyn.extended) return;
        while (!atEnd()) {
            unsigned c = peek();
            if (c == ' ' || (c >= 9 && c <= 13)) {
                ++mPos;
            } else if (c == '#') {
                while (!atEnd() && peek() != '\n') ++mPos;
            } else {
                break;
            }
        }
    }

    NodePtr literal(unsigned c) {
        auto n = mk(Node::Set);
        n->set.add(c);
        if (mSyn.icase) foldCase(n->set);
        return n;
    }

    // ---- zero-width assertions, all built from one-byte look primitives (regex_ast.hpp LookAssert)
    static NodePtr look(bool behind, const ByteSet& set, bool edgeOk) {
        auto n = mk(Node::Assert);
        n->look.behind = behind;
        n->look.set = set;
        n->look.edgeOk = edgeOk;
        return n;
    }
    static NodePtr both(NodePtr a, NodePtr b) {
        auto n = mk(Node::Cat);
        n->kids.push_back(std::move(a));
        n->kids.push_back(std::move(b));
        return n;
    }
    static NodePtr either(NodePtr a, NodePtr b) {
        auto n = mk(Node::Alt);
        n->kids.push_back(std::move(a));
        n->kids.push_back(std::move(b));
        auto g = mk(Node::Group);
        g->kids.push_back(std::move(n));
        return g;
    }
    static ByteSet setOf(std::initializer_list<unsigned> bytes) {
        ByteSet s;
        for (unsigned c : bytes) s.add(c);
        return s;
    }
    static ByteSet complement(ByteSet s) {
        s.invert();
        return s;
    }
    static ByteSet wordSet() {
        ByteSet w;
        shorthand('w', w);
        return w;
    }
    enum class Anchor { BolMulti, BolSingle, EolMulti, EolSingle, WordBoundary, NotWordBoundary, WordStart, WordEnd };
    // Boost semantics: with mod_m, ^ matches at start of input or after \n \r \f, $ at end of input or before them,
    // neither between \r and \n (perl_matcher::match_start_line / match_end_line).
    static NodePtr assertion(Anchor k) {
        const ByteSet none, w = wordSet(), nw = complement(w);
        switch (k) {
            case Anchor::BolSingle: return look(true, none, true);
            case Anchor::EolSingle: return look(false, none, true);
            case Anchor::BolMulti:
                return either(look(true, setOf({'\n', '\f'}), true),
                              both(look(true, setOf({'\r'}), false), look(false, complement(setOf({'\n'})), true)));
            case Anchor::EolMulti:
                return either(look(false, setOf({'\r', '\f'}), true),
                              both(look(false, setOf({'\n'}), false), look(true, complement(setOf({'\r'})), true)));
            case Anchor::WordBoundary:
                return either(both(look(true, w, false), look(false, nw, true)),
                              both(look(true, nw, true), look(false, w, false)));
            case Anchor::NotWordBoundary:
                return either(both(look(true, w, false), look(false, w, false)),
                              both(look(true, nw, true), look(false, nw, true)));
            case Anchor::WordStart: return both(look(true, nw, true), look(false, w, false));
            case Anchor::WordEnd: return both(look(true, w, false), look(false, nw, true));
        }
        return mk(Node::Empty);
    }

    // single-byte escapes shared by atoms and classes; mPos is just past the escape letter.  -1: not one.
    int byteEscape(unsigned letter) {
        switch (letter) {
            case 't': return '\t';
            case 'n': return '\n';
            case 'r': return '\r';
            case 'f': return '\f';
            case 'e': return 27;
            case 'a': return 7;
            case 'x': {
                if (!atEnd() && peek() == '{') {
                    size_t j = mPos + 1;
                    int v = 0, digits = 0;
                    while (j < mPat.size() && hexDigit(static_cast<unsigned char>(mPat[j])) >= 0 && v <= 255) {
                        v = v * 16 + hexDigit(static_cast<unsigned char>(mPat[j]));
                        ++j;
                        ++digits;
                    }
                    if (!digits || j >= mPat.size() || mPat[j] != '}' || v > 255) bail("bad \\x{..} escape");
                    mPos = j + 1;
                    return v;
                }
                int v = 0, digits = 0;
                while (digits < 2 && !atEnd() && hexDigit(peek()) >= 0) {
                    v = v * 16 + hexDigit(peek());
                    ++mPos;
                    ++digits;
                }
                if (!digits) bail("bad \\x escape");
                return v;
            }
            case '0': {
                int v = 0, digits = 0;
                while (digits < 3 && !atEnd() && peek() >= '0' && peek() <= '7') {
                    v = v * 8 + int(peek() - '0');
                    ++mPos;
                    ++digits;
                }
                if (v > 255) bail("octal escape out of range");
                return v;
            }
            case 'c': {
                if (atEnd()) bail("bad \\c escape");
                unsigned v = peek();
                ++mPos;
                return int(v % 32);
            }
            default: return -1;
        }
    }

    NodePtr bracketClass() {  // mPos just past '['
        ByteSet acc;
        bool negate = false;
        if (!atEnd() && peek() == '^') {
            negate = true;
            ++mPos;
        }
        bool first = true;
        for (;;) {
            if (atEnd()) bail("unterminated character class");
            unsigned c = peek();
            if (c == ']' && !first) {
                ++mPos;
                break;
            }
            first = false;
            int lo = -1;
            if (c == '[' && has(1) && (peek(1) == ':' || peek(1) == '=' || peek(1) == '.')) {
                unsigned kind = peek(1);
                size_t j = mPos + 2;
                while (j + 1 < mPat.size() && !(static_cast<unsigned char>(mPat[j]) == kind && mPat[j + 1] == ']')) ++j;
                if (j + 1 >= mPat.size()) bail("unterminated [: :] in class");
                if (kind != ':') bail("collating elements unsupported");
                std::string_view name = mPat.substr(mPos + 2, j - (mPos + 2));
                bool neg = false;
                if (!name.empty() && name[0] == '^') {
                    neg = true;
                    name.remove_prefix(1);
                }
                ByteSet t;
                if (!posixClass(name, t)) bail("unknown POSIX class");
                if (neg) t.invert();
                acc.unite(t);
                mPos = j + 2;
                continue;
            }
            if (c == '\\') {
                ++mPos;
                if (atEnd()) bail("trailing backslash");
                unsigned e = peek();
                ++mPos;
                ByteSet t;
                if (sh(char(e), t)) {
                    acc.unite(t);
                    continue;
                }
                if (e >= 'A' && e <= 'Z' && sh(char(e + 32), t)) {
                    t.invert();
                    acc.unite(t);
                    continue;
                }
                if (e == 'b') {
                    lo = 8;
                } else {
                    int v = byteEscape(e);
                    lo = v >= 0 ? v : int(e);
                }
            } else {
                lo = int(c);
                ++mPos;
            }
            if (has(1) && peek() == '-' && peek(1) != ']') {
                size_t dash = mPos;
                ++mPos;
                unsigned c2 = peek();
                int hi;
                if (c2 == '[' && has(1) && peek(1) == ':') {
                    mPos = dash;
                    acc.add(unsigned(lo));
                    continue;
                }
                if (c2 == '\\') {
                    ++mPos;
                    if (atEnd()) bail("trailing backslash");
                    unsigned e = peek();
                    ++mPos;
                    ByteSet dummy;
                    if (sh(char(e), dummy) || (e >= 'A' && e <= 'Z' && sh(char(e + 32), dummy)))
                        bail("class escape as range endpoint");
                    if (e == 'b') {
                        hi = 8;
                    } else {
                        int v = byteEscape(e);
                        hi = v >= 0 ? v : int(e);
                    }
                } else {
                    hi = int(c2);
                    ++mPos;
                }
                if (hi < lo) bail("invalid range in character class");
                acc.addRange(unsigned(lo), unsigned(hi));
            } else {
                acc.add(unsigned(lo));
            }
        }
        if (mSyn.icase) foldCase(acc);
        if (negate) acc.invert();
        auto n = mk(Node::Set);
        n->set = acc;
        return n;
    }

    // shorthand classes of the active dialect: regexp2's RE2 mode has \d \w as here, \s = [\t\n\f\r ], and no \h \l \u
    // (\v is the VT character there, handled by byteEscape)
    bool sh(char letter, ByteSet& out) const {
        if (!mSyn.regexp2) return shorthand(letter, out);
        if (letter == 's') {
            for (unsigned char c : {'\t', '\n', '\f', '\r', ' '}) out.add(c);
            return true;
        }
        return (letter == 'd' || letter == 'w') && shorthand(letter, out);
    }

    // (?=X) (?!X) (?<=X) (?<!X) with X a single character class: a one-byte look-around.  mPos is just past the
    // introducer.  Longer bodies would need multi-byte look-ahead/behind, which the automata do not have.
    NodePtr lookAround(int depth, bool behind, bool negative) {
        const Syntax saved = mSyn;
        NodePtr body = alternation(depth + 1);
        if (atEnd() || peek() != ')') bail("missing )");
        ++mPos;
        mSyn = saved;
        const Node* n = body.get();
        while ((n->kind == Node::Group && n->capture == 0) && n->kids.size() == 1) n = n->kids[0].get();
        if (!behind && !negative) {  // (?=S*) holds everywhere; (?=(S*)) holds everywhere and captures the run of S
            const Node* g = (n->kind == Node::Group && n->capture && n->kids.size() == 1) ? n : nullptr;
            const Node* r = g ? g->kids[0].get() : n;
            while ((r->kind == Node::Group && r->capture == 0) && r->kids.size() == 1) r = r->kids[0].get();
            if (r->kind == Node::Repeat && r->min == 0 && r->max < 0 && r->greedy) {
                const Node* e = r->kids[0].get();
                while ((e->kind == Node::Group && e->capture == 0) && e->kids.size() == 1) e = e->kids[0].get();
                if (e->kind == Node::Set) {
                    if (!g) return mk(Node::Empty);
                    auto run = mk(Node::Group);
                    run->capture = g->capture;
                    run->runCapture = true;
                    run->set = e->set;
                    run->kids.push_back(mk(Node::Empty));
                    return run;
                }
            }
        }
        if (!behind && n->kind == Node::Cat) {  // (?=AB..) / (?!AB..): decided in sequence() from what follows
            auto a = mk(Node::Assert);
            for (const auto& k : n->kids) {
                const Node* e = k.get();
                while ((e->kind == Node::Group && e->capture == 0) && e->kids.size() == 1) e = e->kids[0].get();
                if (e->kind != Node::Set) return generalLook(std::move(body), false, negative, "unsupported: look-ahead body must be a sequence of character classes");
                a->aheadSeq.push_back(e->set);
            }
            a->aheadNegative = negative;
            return a;
        }
        if (behind && n->kind == Node::Cat) {  // (?<=AB..) / (?<!AB..): decided in sequence() from what stands in front of it
            auto a = mk(Node::Assert);
            for (const auto& k : n->kids) {
                const Node* e = k.get();
                while ((e->kind == Node::Group && e->capture == 0) && e->kids.size() == 1) e = e->kids[0].get();
                if (e->kind != Node::Set) return generalLook(std::move(body), true, negative, "unsupported: look-behind body must be a sequence of character classes");
                a->behindSeq.push_back(e->set);
            }
            a->behindNegative = negative;
            return a;
        }
        if (n->kind != Node::Set) return generalLook(std::move(body), behind, negative, "unsupported: look-around body must be a single character class");
        ByteSet set = n->set;
        if (negative) set.invert();
        return look(behind, set, negative);
    }

    NodePtr group(int depth) {  // mPos just past '('
        Syntax saved = mSyn;
        int capture = 0;
        bool capturing = true;
        std::string name;
        if (!atEnd() && peek() == '?') {
            ++mPos;
            if (atEnd()) bail("unterminated group");
            unsigned d = peek();
            if (d == '#') {
                while (!atEnd() && peek() != ')') ++mPos;
                if (atEnd()) bail("unterminated comment");
                ++mPos;
                return mk(Node::Empty);
            }
            if (d == ':') {
                ++mPos;
                capturing = false;
            } else if (d == '=' || d == '!') {
                ++mPos;
                return lookAround(depth, false, d == '!');
            } else if (d == '>') {  // atomic group: the first way its body matches is final
                ++mPos;
                NodePtr inner = alternation(depth + 1);
                if (atEnd() || peek() != ')') bail("missing )");
                ++mPos;
                mSyn = saved;
                auto a = mk(Node::Atomic);
                a->kids.push_back(std::move(inner));
                return a;
            } else if (d == '(' && has(1) && peek(1) >= '1' && peek(1) <= '9' && !mSyn.namedOnly && !mSyn.regexp2) {
                // (?(N)yes|no): yes when group N has taken part, else no (absent: nothing).  Not regular bookkeeping: the device
                // backtracking engine (bt_vm.hpp BT_COND).  Conditions on names, look-arounds or recursion stay refused.
                ++mPos;
                int v = 0;
                while (!atEnd() && peek() >= '0' && peek() <= '9' && v < 1000) v = v * 10 + int(peek() - '0'), ++mPos;
                if (atEnd() || peek() != ')') bail("unsupported group construct (recursion/conditional)");
                ++mPos;
                NodePtr inner = alternation(depth + 1);
                if (atEnd() || peek() != ')') bail("missing )");
                ++mPos;
                mSyn = saved;
                auto c = mk(Node::Cond);
                c->capture = v;
                if (inner->kind == Node::Alt) {
                    if (inner->kids.size() > 2) bail("a conditional has at most two alternatives");
                    c->kids.push_back(std::move(inner->kids[0]));
                    c->kids.push_back(std::move(inner->kids[1]));
                } else {
                    c->kids.push_back(std::move(inner));
                    c->kids.push_back(mk(Node::Empty));
                }
                mBackRefs.push_back(v);   // (validated like a reference: the group has to exist)
                mGeneralLook = true;
                return c;
            } else if (d == '|' || d == '(' || d == 'R' || d == '&' || d == '+' || (d >= '0' && d <= '9')) {
                bail("unsupported group construct (recursion/conditional)");
            } else if (d == '<' || d == 'P' || d == '\'') {
                char close = '>';
                if (d == 'P') {
                    ++mPos;
                    if (atEnd() || peek() != '<') bail("unsupported (?P construct");
                } else if (d == '\'') {
                    close = '\'';
                } else if (has(1) && (peek(1) == '=' || peek(1) == '!')) {
                    const bool negative = peek(1) == '!';
                    mPos += 2;
                    return lookAround(depth, true, negative);
                }
                ++mPos;
                size_t j = mPos;
                while (j < mPat.size() && mPat[j] != close) ++j;
                if (j >= mPat.size() || j == mPos) bail("bad group name");
                name = std::string(mPat.substr(mPos, j - mPos));
                mPos = j + 1;
            } else {
                // inline option letters
                Syntax next = mSyn;
                bool on = true;
                for (;; ++mPos) {
                    if (atEnd()) bail("unterminated flag group");
                    d = peek();
                    if (d == '-') {
                        on = false;
                    } else if (d == 'i') {
                        next.icase = on;
                    } else if (d == 's') {
                        next.dotAll = on;
                    } else if (d == 'm') {
                        next.multiLine = on;
                    } else if (d == 'x') {
                        next.extended = on;
                    } else {
                        break;
                    }
                }
                if (d == ')') {
                    ++mPos;
                    mSyn = next;  // applies to the remainder of the enclosing group
                    return mk(Node::Empty);
                }
                if (d != ':') bail("unknown inline flag");
                ++mPos;
                mSyn = next;
                capturing = false;
            }
        }
        if (capturing && name.empty() && mSyn.namedOnly) capturing = false;
        if (capturing) {
            if (mGroups >= 255) bail("too many capture groups");
            capture = ++mGroups;
            mNames.push_back(name);
        }
        NodePtr inner = alternation(depth + 1);
        if (atEnd() || peek() != ')') bail("missing )");
        ++mPos;
        mSyn = saved;
        auto g = mk(Node::Group);
        g->capture = capture;
        g->kids.push_back(std::move(inner));
        return g;
    }

    NodePtr escapeAtom() {  // mPos just past '\'
        if (atEnd()) bail("trailing backslash");
        unsigned e = peek();
        ++mPos;
        {
            ByteSet t;
            if (sh(char(e), t)) {
                if (mSyn.icase) foldCase(t);
                auto n = mk(Node::Set);
                n->set = t;
                return n;
            }
            if (e >= 'A' && e <= 'Z' && sh(char(e + 32), t)) {
                if (mSyn.icase) foldCase(t);
                t.invert();
                auto n = mk(Node::Set);
                n->set = t;
                return n;
            }
        }
        switch (e) {
            case 'b': return assertion(Anchor::WordBoundary);
            case 'B': return assertion(Anchor::NotWordBoundary);
            case '<':
                if (mSyn.regexp2) return literal(e);
                return assertion(Anchor::WordStart);
            case '>':
                if (mSyn.regexp2) return literal(e);
                return assertion(Anchor::WordEnd);
            case '`':
                if (mSyn.regexp2) return literal(e);
                return assertion(Anchor::BolSingle);
            case 'A': return assertion(Anchor::BolSingle);
            case '\'':
                if (mSyn.regexp2) return literal(e);
                return assertion(Anchor::EolSingle);
            case 'z': return assertion(Anchor::EolSingle);
            case 'Z': bail("\\Z (multi-byte look-ahead) unsupported");
            case 'Q': {
                auto seq = mk(Node::Cat);
                while (!atEnd()) {
                    if (peek() == '\\' && has(1) && peek(1) == 'E') {
                        mPos += 2;
                        break;
                    }
                    seq->kids.push_back(literal(peek()));
                    ++mPos;
                }
                auto g = mk(Node::Group);
                g->kids.push_back(std::move(seq));
                return g;
            }
            case 'E': return mk(Node::Empty);
            case 'k': case 'g': {
                // \k<name> \k{name} \k'name' \g{name}: a back-reference by name; \gN \g{N} \g{-N}: by number / relative to here (boost
                // Perl syntax).  Only to groups opened in front of the reference; not under Grok's dialect (see \N below).
                // Grok (regexp2, named-only numbering here): by NAME only -- regexp2 numbers unnamed groups first, so a number written
                // in a Grok pattern does not name the group it names there.  The Go regex plugin (RE2): none at all.
                if ((mSyn.regexp2 && !mSyn.namedOnly) || atEnd()) bail("unsupported escape");
                if (mSyn.namedOnly && (e != 'k' || (peek() != '<' && peek() != '{' && peek() != '\''))) bail("unsupported escape");
                const unsigned open = peek();
                const unsigned close = open == '<' ? '>' : open == '{' ? '}' : open == '\'' ? '\'' : 0;
                std::string word;
                if (close) {
                    ++mPos;
                    while (!atEnd() && peek() != close) word.push_back(char(peek())), ++mPos;
                    if (atEnd() || word.empty()) bail("unterminated back-reference name");
                    ++mPos;
                } else if (e == 'g') {
                    if (peek() == '-') word.push_back('-'), ++mPos;
                    while (!atEnd() && peek() >= '0' && peek() <= '9') word.push_back(char(peek())), ++mPos;
                    if (word.empty() || word == "-") bail("unsupported escape");
                } else {
                    bail("unsupported escape");
                }
                int v = 0;
                const bool numeric = word.find_first_not_of("-0123456789") == std::string::npos;
                if (numeric && (e == 'g')) {
                    const int k = std::atoi(word.c_str());
                    v = k < 0 ? mGroups + 1 + k : k;
                    if (k == 0 || v < 1) bail("invalid back reference");
                } else {
                    for (int g = 1; g <= mGroups && !v; ++g)
                        if (mNames[size_t(g)] == word) v = g;
                    if (!v) bail("back-reference to a group name that is not defined in front of it: unsupported");
                    mNamedRefs.push_back(word);
                }
                auto n = mk(Node::BackRef);
                n->capture = v;
                mBackRefs.push_back(v);
                return n;
            }
            case 'p': case 'P': case 'X': case 'C': case 'R': case 'K': case 'G': case 'N':
                bail("unsupported escape");
            case 'u':
                if (mSyn.regexp2) bail("unsupported escape");  // \uXXXX
                break;
            default: break;
        }
        if (e >= '1' && e <= '9') {
            // \N: a back-reference (boost perl_matcher::match_backref; regexp2 likewise): the bytes group N matched last, again.  Not
            // regular: the pattern goes to the device backtracking engine (bt_vm.hpp).  Under Grok's named-only numbering the digits
            // would not count the groups the author sees: refused there.
            // Nor in the Go plugins' escape dialect: processor_regex compiles with Go's regexp (RE2), which has no back-references.
            if (mSyn.namedOnly || mSyn.regexp2) bail("back-references with named-only captures / in the Go plugins' dialect unsupported");
            unsigned v = e - '0';
            while (has(0) && peek() >= '0' && peek() <= '9' && v * 10 + (peek() - '0') <= 999) {
                v = v * 10 + (peek() - '0');
                ++mPos;
            }
            auto n = mk(Node::BackRef);
            n->capture = int(v);
            mBackRefs.push_back(int(v));
            return n;
        }
        int v = byteEscape(e);
        return literal(v >= 0 ? unsigned(v) : e);
    }

    // returns nullptr when the next token cannot start an atom
    NodePtr atom(int depth, bool& isAssertion) {
        isAssertion = false;
        unsigned c = peek();
        switch (c) {
            case '(': ++mPos; return group(depth);
            case '[': ++mPos; return bracketClass();
            case '.': {
                ++mPos;
                auto n = mk(Node::Set);
                n->set = ByteSet::all();
                if (!mSyn.dotAll) n->set.w[0] &= ~(uint64_t(1) << '\n');
                return n;
            }
            case '^':
                ++mPos;
                isAssertion = true;
                return assertion(mSyn.multiLine ? Anchor::BolMulti : Anchor::BolSingle);
            case '$':
                ++mPos;
                isAssertion = true;
                return assertion(mSyn.multiLine ? Anchor::EolMulti : Anchor::EolSingle);
            case '*': case '+': case '?': bail("nothing to repeat");
            case '\\': {
                ++mPos;
                const unsigned e = has(0) ? peek() : 0;
                NodePtr n = escapeAtom();
                isAssertion = std::strchr(mSyn.regexp2 ? "bBAzZ" : "bB<>AzZ`'", int(e)) != nullptr && e != 0;
                return n;
            }
            default: ++mPos; return literal(c);
        }
    }

    bool counted(int& lo, int& hi) {  // at '{'
        size_t j = mPos + 1;
        long a = 0, b = -1;
        int digits = 0;
        while (j < mPat.size() && mPat[j] >= '0' && mPat[j] <= '9') {
            a = a * 10 + (mPat[j] - '0');
            if (a > 100000) return false;
            ++j;
            ++digits;
        }
        if (!digits || j >= mPat.size()) return false;
        if (mPat[j] == '}') {
            b = a;
            ++j;
        } else if (mPat[j] == ',') {
            ++j;
            long t = 0;
            digits = 0;
            while (j < mPat.size() && mPat[j] >= '0' && mPat[j] <= '9') {
                t = t * 10 + (mPat[j] - '0');
                if (t > 100000) return false;
                ++j;
                ++digits;
            }
            if (j >= mPat.size() || mPat[j] != '}') return false;
            ++j;
            b = digits ? t : -1;
        } else {
            return false;
        }
        lo = int(a);
        hi = int(b);
        mPos = j;
        return true;
    }

    NodePtr sequence(int depth) {
        auto seq = mk(Node::Cat);
        for (;;) {
            skipFreeSpacing();
            if (atEnd() || peek() == '|' || peek() == ')') break;
            bool isAssertion = false;
            NodePtr a = atom(depth, isAssertion);
            bool quantified = false;
            for (;;) {
                skipFreeSpacing();
                if (atEnd()) break;
                unsigned q = peek();
                int lo, hi;
                if (q == '*') {
                    lo = 0; hi = -1; ++mPos;
                } else if (q == '+') {
                    lo = 1; hi = -1; ++mPos;
                } else if (q == '?') {
                    lo = 0; hi = 1; ++mPos;
                } else if (q == '{') {
                    if (!counted(lo, hi)) break;
                } else {
                    break;
                }
                if (quantified) bail("nested quantifier");
                if (hi >= 0 && hi < lo) bail("bad repeat range");
                bool greedy = true;
                if (!atEnd() && peek() == '?') {
                    greedy = false;
                    ++mPos;
                }
                bool possessive = false;
                if (greedy && !atEnd() && peek() == '+') {  // X*+  ==  (?>X*)
                    possessive = true;
                    ++mPos;
                }
                auto r = mk(Node::Repeat);
                r->min = lo;
                r->max = hi;
                r->greedy = greedy;
                r->kids.push_back(std::move(a));
                a = std::move(r);
                if (possessive) {
                    auto at = mk(Node::Atomic);
                    at->kids.push_back(std::move(a));
                    a = std::move(at);
                }
                quantified = true;
            }
            seq->kids.push_back(std::move(a));
        }
        // Multi-byte look-aheads (the library's "(?!<[0-9])%{HOUR}" -- a mistyped look-behind): the automata only look
        // one byte ahead, but when the rest of this sequence cannot begin with the look-ahead's first class the answer
        // is already known on every path that goes on to match: a negative one holds, a positive one fails.  What the rest of the
        // sequence does not decide stays in the tree as a window (regex_ast.hpp; follow_nfa.cpp builds the product).
        for (size_t i = 0; i < seq->kids.size(); ++i) {
            Node& k = *seq->kids[i];
            if (k.kind != Node::Assert || k.aheadSeq.empty() || k.window) continue;
            ByteSet first;
            bool restNullable = true;
            for (size_t j = i + 1; j < seq->kids.size() && restNullable; ++j) restNullable = firstOf(*seq->kids[j], first);
            bool disjoint = !restNullable;
            for (int w = 0; w < 4 && disjoint; ++w) disjoint = (first.w[w] & k.aheadSeq[0].w[w]) == 0;
            if (!disjoint) {
                k.window = true;
                continue;
            }
            if (k.aheadNegative) {
                seq->kids[i] = mk(Node::Empty);
            } else {
                auto never = mk(Node::Set);  // empty class: this branch cannot match
                seq->kids[i] = std::move(never);
            }
        }
        // Multi-byte look-behinds ("\\{ (?<={ )" of the library's MONGO_QUERY): decided here when the sub-expressions in front of the
        // assertion, inside this sequence, have a fixed width that covers the body -- the bytes the body looks at are then bytes this
        // sequence has just consumed, position by position.  Body class j holds for certain when it contains every byte the pattern
        // allows there, and fails for certain when it contains none of them.  Anything else (the body reaches back beyond this sequence,
        // or a class overlaps without containing) would need the history of the input in the automaton: refused.
        for (size_t i = 0; i < seq->kids.size(); ++i) {
            Node& k = *seq->kids[i];
            if (k.kind != Node::Assert || k.behindSeq.empty()) continue;
            std::vector<ByteSet> before;  // before[d] = bytes the pattern allows d + 1 bytes in front of the assertion
            for (size_t j = i; j-- > 0 && before.size() < k.behindSeq.size();) {
                std::vector<ByteSet> w;
                if (!fixedWidthSets(*seq->kids[j], w)) break;
                for (size_t q = w.size(); q-- > 0;) before.push_back(w[q]);
            }
            auto asGeneral = [&]() {  // the class sequence as a body the backtracking engine runs k bytes back
                auto body = mk(Node::Cat);
                for (const ByteSet& b : k.behindSeq) {
                    auto e = mk(Node::Set);
                    e->set = b;
                    body->kids.push_back(std::move(e));
                }
                seq->kids[i] = generalLook(std::move(body), true, k.behindNegative,
                                           "unsupported: multi-byte look-behind that the preceding sub-expression does not decide");
            };
            if (before.size() < k.behindSeq.size()) {
                asGeneral();
                continue;
            }
            bool holds = true, fails = false;
            const size_t n = k.behindSeq.size();
            for (size_t d = 0; d < n; ++d) {
                const ByteSet& want = k.behindSeq[n - 1 - d];
                const ByteSet& have = before[d];
                bool subset = true, meet = false;
                for (int w = 0; w < 4; ++w) {
                    subset = subset && (have.w[w] & ~want.w[w]) == 0;
                    meet = meet || (have.w[w] & want.w[w]) != 0;
                }
                holds = holds && subset;
                fails = fails || !meet;
            }
            if (!holds && !fails) {
                asGeneral();
                continue;
            }
            const bool truth = k.behindNegative ? fails : holds;
            if (truth) seq->kids[i] = mk(Node::Empty);
            else seq->kids[i] = mk(Node::Set);  // empty class: this branch cannot match
        }
        if (seq->kids.empty()) return mk(Node::Empty);
        if (seq->kids.size() == 1) {
            if (seq->kids[0]->kind == Node::Assert && !seq->kids[0]->aheadSeq.empty()) seq->kids[0]->window = true;
            return std::move(seq->kids[0]);
        }
        return seq;
    }

    // the byte classes of a sub-expression every match of which has the same length (appended to `out` in input order); false if it has
    // no such form (a repeat with a range, alternatives of different shapes, assertions inside)
    static bool fixedWidthSets(const Node& n, std::vector<ByteSet>& out) {
        switch (n.kind) {
            case Node::Empty: return true;
            case Node::Set: out.push_back(n.set); return true;
            case Node::Cat:
                for (const auto& k : n.kids)
                    if (!fixedWidthSets(*k, out)) return false;
                return true;
            case Node::Group:
                if (n.runCapture) return false;
                return fixedWidthSets(*n.kids[0], out);
            case Node::Repeat: {
                if (n.min != n.max || n.min < 0 || n.min > 64) return false;
                for (int r = 0; r < n.min; ++r)
                    if (!fixedWidthSets(*n.kids[0], out)) return false;
                return true;
            }
            case Node::Alt: {  // alternatives of one width: position by position the union
                std::vector<ByteSet> acc;
                for (size_t a = 0; a < n.kids.size(); ++a) {
                    std::vector<ByteSet> w;
                    if (!fixedWidthSets(*n.kids[a], w)) return false;
                    if (a == 0) acc = w;
                    else {
                        if (w.size() != acc.size()) return false;
                        for (size_t q = 0; q < w.size(); ++q) acc[q].unite(w[q]);
                    }
                }
                out.insert(out.end(), acc.begin(), acc.end());
                return true;
            }
            case Node::Assert: return n.aheadSeq.empty() && n.behindSeq.empty();  // (one-byte assertions are zero-width: transparent)
            case Node::Look: return true;  // (zero-width)
            case Node::Atomic: return false;
        }
        return false;
    }

    // FIRST set of a sub-expression (bytes a match of it can begin with) ORed into `out`; returns whether it can match
    // the empty string.  Zero-width assertions are transparent.
    static bool firstOf(const Node& n, ByteSet& out) {
        switch (n.kind) {
            case Node::Empty:
            case Node::Look:
            case Node::Assert: return true;
            case Node::Set:
                for (int w = 0; w < 4; ++w) out.w[w] |= n.set.w[w];
                return false;
            case Node::Cat:
                for (const auto& k : n.kids)
                    if (!firstOf(*k, out)) return false;
                return true;
            case Node::Alt: {
                bool any = false;
                for (const auto& k : n.kids) any |= firstOf(*k, out);
                return any;
            }
            case Node::Repeat: {
                const bool inner = firstOf(*n.kids[0], out);
                return inner || n.min == 0;
            }
            case Node::Group:
            case Node::Atomic: return firstOf(*n.kids[0], out);
            case Node::BackRef:  // (whatever the group held, the empty string included)
                for (int w = 0; w < 4; ++w) out.w[w] = ~uint64_t(0);
                return true;
            case Node::Cond: {
                const bool a = firstOf(*n.kids[0], out), b = firstOf(*n.kids[1], out);
                return a || b;
            }
        }
        return true;
    }

    NodePtr alternation(int depth) {
        if (depth > 200) bail("nesting too deep");
        NodePtr first = sequence(depth);
        if (atEnd() || peek() != '|') return first;
        auto alt = mk(Node::Alt);
        alt->kids.push_back(std::move(first));
        while (!atEnd() && peek() == '|') {
            ++mPos;
            alt->kids.push_back(sequence(depth));
        }
        return alt;
    }
};

}  // namespace

ParsedRegex parseRegex(std::string_view pattern, Syntax syntax) { return Parser(pattern, syntax).run(); }

}  // namespace lcregex
