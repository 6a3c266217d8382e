#!/usr/bin/env python3
"""Golden vectors for back-references \\1 .. \\N (round 6: the device backtracking engine, csrc/bt_vm.hpp).

boost::regex (Perl syntax, what core/common/StringTools.cpp:183-211 calls) is on neither box, so these vectors are the agreement of
THREE independent backtracking engines that give back-references Perl's meaning: CPython `re`, the `regex` module (both bytes,
DOTALL|MULTILINE) and PCRE1 8.45 (the wrapper of gen_regex_golden.py).  A pattern CPython cannot compile (atomic groups, possessive
quantifiers: 3.10) is pinned by the other two.  Vectors on which the engines disagree are dropped and counted.

Left out on purpose: a group that refers to ITSELF from inside ("(a\\1?)"), where Perl-family engines differ in what the half-open
group holds.

Named and relative references (\\k<name>, \\g{-1} ...) are read by PCRE1 alone of the three: each such pattern is listed beside its \\N
spelling and must give what all three engines give for that ("named_full").

Writes tests/golden/backref_vectors.json: {"full": [...], "search": [...], "icase_full": [...], "named_full": [...]}, each list of
{p, g, subs: [[subject, flat caps incl. group 0 or null], ...]} (latin-1 strings).
Run from the repo root:  python tests/golden/gen_backref_golden.py
"""
import json
import os
import random
import re
import sys

import regex

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_regex_golden import Pcre  # noqa: E402

CURATED = [
    (rb'(\w+) \1', [b'abc abc', b'abc abd', b'a a', b'ab abab', b' ']),
    (rb'(a*)b\1', [b'aabaa', b'b', b'aaba', b'abaa', b'ab']),
    (rb'(["\'])(.*?)\1', [b'"x"', b"'it'", b'"a\'', b'""', b'"a"b"']),
    (rb'(["\'])((?:\\.|[^\\])*?)\1 (\d+)', [b'"a b" 12', b"'q\\'r' 3", b'"a\' 1', b'"" 0']),
    (rb'<(\w+)>(.*)</\1>', [b'<a>x</a>', b'<a>x</b>', b'<ab><ab>y</ab></ab>', b'<a></a>']),
    (rb'(\d\d)-\1-\1', [b'12-12-12', b'12-12-13', b'1-1-1']),
    (rb'(a|b)\1+', [b'aaa', b'bb', b'ab', b'a']),
    (rb'(a)|(b)\2', [b'a', b'bb', b'b']),
    (rb'(?:(a)|b)\1', [b'aa', b'b', b'ba']),
    (rb'(a)?\1b', [b'aab', b'b', b'ab']),
    (rb'(\w)(\w)\2\1', [b'abba', b'abab', b'aaaa', b'abb']),
    (rb'(.)\1', [b'aa', b'ab', b'\n\n']),
    (rb'(.*)\1', [b'abab', b'aa', b'', b'aba']),
    (rb'(.+)\1', [b'abab', b'aaaa', b'a', b'abcabc']),
    (rb'(.+?)\1(.*)', [b'ababab', b'aaaa', b'xyxyz']),
    (rb'(a+)(b+)\2\1', [b'aabbbbaa', b'abba', b'abab', b'aabbbaa']),
    (rb'(\w+)=(\w+);\1=\2', [b'k=v;k=v', b'k=v;k=w', b'ab=c;ab=c']),
    (rb'(x)(y)(z)(a)(b)(c)(d)(e)(f)(g)\10', [b'xyzabcdefgg', b'xyzabcdefgx']),
    (rb'(\S+) (\S+) \2 \1', [b'GET /x /x GET', b'a b b a', b'a b a b']),
    (rb'^(\w+)\s+\1$', [b'hi hi', b'hi  hi', b'hi ho']),
    (rb'(\d+)\.(\d+)\.\1\.\2', [b'10.20.10.20', b'10.20.10.2', b'1.2.1.2']),
    (rb'((a)|(b))+\2', [b'aba', b'aa', b'ab', b'bba']),
    (rb'(?:(\w)\1)+', [b'aabb', b'aab', b'aa', b'abab']),
    (rb'(\w+?)\1*', [b'ababab', b'aaa', b'abc']),
    (rb'(a*?)\1b', [b'aab', b'b', b'aaab']),
    (rb'\b(\w+)\b.*\b\1\b', [b'the cat the', b'the cat then', b'a b a']),
    (rb'(\[)?\w+(?:\])?\1?', [b'[ab]', b'ab', b'[ab][']),
    (rb'(?>(a+))\1', [b'aaaa', b'aa', b'a']),
    (rb'(a++)b\1', [b'aabaa', b'aaba']),
    (rb'(\w+)@\1\.(com|org)', [b'foo@foo.com', b'foo@bar.com', b'x@x.org', b'x@x.net']),
    (rb'([0-9a-f]{2})(?::\1){2}', [b'ab:ab:ab', b'ab:ab:ac', b'00:00:00']),
]


# named and relative forms (boost Perl syntax: \k<name> \k{name} \k'name' \g{name} \g{N} \gN \g{-N}).  Of the three engines only PCRE1
# reads them all; each is listed beside its \N spelling and must give what the three engines give for THAT.
NAMED = [
    (rb'(?<q>["\'])(.*?)\k<q>', rb'(["\'])(.*?)\1', [b'"x"', b"'it'", b'"a\'', b'""', b'"a"b"']),
    (rb"(?<w>\w+) \k{w}", rb"(\w+) \1", [b"abc abc", b"abc abd", b"a a", b"ab abab"]),
    (rb"(?<w>\w+) \k'w'", rb"(\w+) \1", [b"abc abc", b"abc abd"]),
    (rb"(?<w>\w+) \g{w}", rb"(\w+) \1", [b"abc abc", b"abc abd"]),
    (rb"(\d\d)-\g{1}-\g1", rb"(\d\d)-\1-\1", [b"12-12-12", b"12-12-13", b"1-1-1"]),
    (rb"(a)(?<x>b+)\g{-1}\g{-2}", rb"(a)(b+)\2\1", [b"abba", b"abbbba", b"abb", b"abab"]),
    (rb"(\w)(\w)\g{-1}\g{-2}", rb"(\w)(\w)\2\1", [b"abba", b"abab", b"aaaa"]),
    (rb"<(?<tag>\w+)>(.*)</\k<tag>>", rb"<(\w+)>(.*)</\1>", [b"<a>x</a>", b"<a>x</b>", b"<ab><ab>y</ab></ab>"]),
    (rb"(?<k>\w+)=(?<v>\w+);\k<k>=\k<v>", rb"(\w+)=(\w+);\1=\2", [b"k=v;k=v", b"k=v;k=w", b"ab=c;ab=c"]),
]


# general look-arounds (round 6, the same engine): bodies that are not a byte class or a sequence of byte classes -- repeats, alternations,
# captures, back-references inside; look-behinds of fixed length.  All three engines read them.
LOOK = [
    (rb'(?=a+b)(\w+)', [b'aab', b'aaab1', b'aac', b'b']),
    (rb'(\w+?)(?=\d+$)(\d+)', [b'abc123', b'a1', b'abc', b'12']),
    (rb'(?!\d+x)(\w+)', [b'12x', b'12y', b'x', b'1x2']),
    (rb'(\w+) (?=(\w+) \2)(.*)', [b'a b b', b'a b c', b'k vv vv z']),
    (rb'(\w+)(?= \1) .*', [b'ab ab', b'ab abc', b'ab cd']),
    (rb'(?=(a+))a*b\1', [b'aaba', b'aabaa', b'ab', b'aab']),
    (rb'(.*)(?<=ab|cd)(x*)', [b'abx', b'cdxx', b'acx', b'ab', b'zzcd']),
    (rb'(\w+)(?<!\d\d)-(\w+)', [b'ab-c', b'a12-c', b'a1-c', b'12-3']),
    (rb'(?<=(?:ab){2})c(.*)', [b'ababc', b'abc']),
    (rb'(.*?)(?<=[a-c]{2}\d)!(.*)', [b'ab1!x', b'ad1!x', b'zzbc9!', b'c9!']),
    (rb'((?:(?!<b>).)*)<b>(.*)', [b'abc<b>x', b'<b>', b'a<b<b>y', b'abc']),
    (rb'(?:(?=\w+:)(\w+):|(\d+))+', [b'k:12', b'12k:', b'a:b:', b'a:b']),
    (rb'(?=.*\d)(?=.*[a-z])(\S{4,})', [b'abc1', b'abcd', b'1234', b'a1', b'xx9yy']),
    (rb'(a|ab)(?=c|bc)(.*)', [b'abc', b'ac', b'abbc', b'ab']),
    (rb'(?!(a)\1)(\w+)', [b'aab', b'aba', b'ab']),
    (rb'(\d+)(?<=[13579])(\w*)', [b'123x', b'124x', b'9', b'12']),
    (rb'(?<=,|;)(\w+)(?=,|$)(.*)', [b',ab,cd', b';ab', b',ab cd', b'ab']),
]


# conditionals on a group (?(N)yes|no): all three engines read them
COND = [
    (rb'(\()?\w+(?(1)\))', [b'(ab)', b'ab', b'(ab', b'ab)']),
    (rb'(<)?(\w+)(?(1)>|;)(.*)', [b'<a>x', b'a;x', b'<a;x', b'a>x']),
    (rb'^(?:(a)|b)(?(1)c|d)$', [b'ac', b'bd', b'ad', b'bc']),
    (rb'(")?(\w+)(?(1)")=(\d+)', [b'"k"=1', b'k=1', b'"k=1', b'k"=1']),
    (rb'(a)?(b)?(?(1)x|y)(?(2)z)', [b'abxz', b'ax', b'byz', b'y', b'bxz']),
    (rb'(?:(\d+)|(\w+))-(?(1)\d|\w)+', [b'12-34', b'ab-cd', b'12-ab', b'ab-12']),
]


def engines_full(p, s, flags_re, flags_rx, pcre, icase):
    r2 = regex.compile(p, flags_rx)
    m2 = r2.fullmatch(s)
    ng = r2.groups
    e2 = None if m2 is None else [list(m2.span(g)) for g in range(ng + 1)]
    outs = [e2]
    try:
        r1 = re.compile(p, flags_re)
        m1 = r1.fullmatch(s)
        outs.append(None if m1 is None else [list(m1.span(g)) for g in range(ng + 1)])
    except re.error:
        pass
    if not icase:
        outs.append(pcre.fullmatch(p, s, ng))
    return outs, ng


def engines_search(p, s, pcre):
    r2 = regex.compile(p, regex.S | regex.M)
    m2 = r2.search(s)
    ng = r2.groups
    outs = [None if m2 is None else [list(m2.span(g)) for g in range(ng + 1)]]
    try:
        r1 = re.compile(p, re.S | re.M)
        m1 = r1.search(s)
        outs.append(None if m1 is None else [list(m1.span(g)) for g in range(ng + 1)])
    except re.error:
        pass
    outs.append(pcre.search(p, s, ng))
    return outs, ng


def gen(rng, groups, d=0):
    """random pattern text; groups = [count of capture groups CLOSED so far] (a reference only names a closed group)"""
    r = rng.random()
    if r < 0.30 or d > 2:
        return rng.choice(['a', 'b', 'c', '[ab]', '[^a]', '.', r'\d', r'\w'])
    if r < 0.42 and groups[0] > 0:
        return '\\%d' % rng.randint(1, groups[0])
    if r < 0.56:
        return gen(rng, groups, d + 1) + gen(rng, groups, d + 1)
    if r < 0.66:
        return '(?:' + gen(rng, groups, d + 1) + '|' + gen(rng, groups, d + 1) + ')'
    if r < 0.84:
        inner = gen(rng, groups, d + 1)
        groups[0] += 1
        return '(' + inner + ')'
    if r < 0.87:
        return rng.choice([r'\b', '^', '$'])
    q = rng.choice(['*', '+', '?', '{1,2}', '*?', '+?', '??'])
    return '(?:' + gen(rng, groups, d + 1) + ')' + q


def main():
    rng = random.Random(20260930)
    pcre = Pcre()
    out = {"full": [], "search": [], "icase_full": []}
    dropped = 0

    def add(kind, p, s):
        nonlocal dropped
        try:
            if kind in ("full", "look_full", "cond_full"):
                outs, ng = engines_full(p, s, re.S | re.M, regex.S | regex.M, pcre, False)
            elif kind == "icase_full":
                outs, ng = engines_full(p, s, re.S | re.M | re.I, regex.S | regex.M | regex.I, pcre, True)
            else:
                outs, ng = engines_search(p, s, pcre)
        except (regex.error, ValueError):
            return
        if len(outs) < 2 or any(o != outs[0] for o in outs[1:]):
            dropped += 1
            return
        exp = outs[0]
        lst = out[kind]
        ent = next((c for c in lst if c["p"] == p.decode("latin-1")), None)
        if ent is None:
            ent = {"p": p.decode("latin-1"), "g": ng, "subs": []}
            lst.append(ent)
        flat = None if exp is None else [v for ab in exp for v in ab]
        rec = [s.decode("latin-1"), flat]
        if rec not in ent["subs"]:
            ent["subs"].append(rec)

    for p, subs in CURATED:
        for s in subs:
            add("full", p, s)
            add("search", p, b"zz " + s + b" !")
            add("search", p, s)
    for p, subs in [(rb'(\w+) \1', [b'abc ABC', b'Abc aBC', b'abc abd']), (rb'([a-c])x\1', [b'axA', b'BxB', b'axb']),
                    (rb'(k)=(v) \2\1', [b'K=v VK', b'k=V vk', b'k=v kv'])]:
        for s in subs:
            add("icase_full", p, s)
    out["look_full"], out["look_search"] = [], []
    for p, subs in LOOK:
        for s_ in subs:
            add("look_full", p, s_)
            add("look_search", p, b"zz " + s_ + b" !")
            add("look_search", p, s_)
    out["cond_full"] = []
    out["cond_search"] = []
    for p, subs in COND:
        for s_ in subs:
            add("cond_full", p, s_)
            add("cond_search", p, b"zz " + s_ + b" !")
            add("cond_search", p, s_)
    out["named_full"] = []
    for named, numeric, subs in NAMED:
        for s_ in subs:
            outs, ng = engines_full(numeric, s_, re.S | re.M, regex.S | regex.M, pcre, False)
            assert len(outs) == 3 and all(o == outs[0] for o in outs), (numeric, s_)
            exp = pcre.fullmatch(named, s_, ng)
            assert exp == outs[0], (named, s_, exp, outs[0])
            ent = next((c for c in out["named_full"] if c["p"] == named.decode("latin-1")), None)
            if ent is None:
                ent = {"p": named.decode("latin-1"), "g": ng, "numeric": numeric.decode("latin-1"), "subs": []}
                out["named_full"].append(ent)
            ent["subs"].append([s_.decode("latin-1"), None if exp is None else [v for ab in exp for v in ab]])
    n = 0
    while n < 500:
        groups = [0]
        p = gen(rng, groups).encode()
        if b'\\1' not in p and b'\\2' not in p and b'\\3' not in p:
            continue
        try:
            regex.compile(p)
        except regex.error:
            continue
        n += 1
        for _ in range(8):
            s = bytes(rng.choice(b'abc1 ') for _ in range(rng.randint(0, 8)))
            add("full", p, s)
        for _ in range(3):
            s = bytes(rng.choice(b'abc1 ') for _ in range(rng.randint(0, 10)))
            add("search", p, s)
    res = {"generator": "tests/golden/gen_backref_golden.py", "seed": 20260930,
           "engines": ["CPython re %s (bytes, DOTALL|MULTILINE)" % sys.version.split()[0],
                       "regex %s (bytes, DOTALL|MULTILINE)" % regex.__version__, "PCRE1 8.45 (DOTALL|MULTILINE; not the icase vectors)"],
           "dropped_disagreements": dropped,
           "n_full": sum(len(c["subs"]) for c in out["full"]), "n_search": sum(len(c["subs"]) for c in out["search"]),
           "n_icase_full": sum(len(c["subs"]) for c in out["icase_full"]),
           "n_named_full": sum(len(c["subs"]) for c in out["named_full"]),
           "n_cond_full": sum(len(c["subs"]) for c in out["cond_full"]), "n_cond_search": sum(len(c["subs"]) for c in out["cond_search"]),
           "n_look_full": sum(len(c["subs"]) for c in out["look_full"]), "n_look_search": sum(len(c["subs"]) for c in out["look_search"]),
           "format": "full/search/icase_full[i] = {p, g, subs: [[subject, flat caps incl. group 0 or null], ...]}",
           "full": out["full"], "search": out["search"], "icase_full": out["icase_full"], "named_full": out["named_full"],
           "look_full": out["look_full"], "look_search": out["look_search"], "cond_full": out["cond_full"],
           "cond_search": out["cond_search"]}
    with open(os.path.join(HERE, "backref_vectors.json"), "w") as f:
        json.dump(res, f, separators=(",", ":"))
    print("full", res["n_full"], "search", res["n_search"], "icase", res["n_icase_full"], "dropped", dropped,
          "patterns", len(out["full"]), len(out["search"]))


if __name__ == "__main__":
    main()
