#!/usr/bin/env python3
"""Golden vectors for atomic groups (?>X) and possessive quantifiers X*+ X++ X?+ X{m,n}+.

CPython 3.10 `re` has neither, so these vectors are the agreement of two OTHER independent backtracking engines:
the `regex` module (bytes, DOTALL|MULTILINE) and PCRE1 8.45 (the same wrapper as gen_regex_golden.py).  boost::regex
(Perl syntax, what core/common/StringTools.cpp:183-211 calls) and Oniguruma/regexp2-style Grok patterns
(plugins/processor/grok/processor_grok_default_patterns.go: BASE10NUM, YEAR, QUOTEDSTRING ...) give these constructs
the same meaning: once X has matched, no backtracking into X.

Writes tests/golden/regex_atomic_golden.json: {"full": [...], "search": [...]}, each list of
{p, g, subs: [[subject, flat caps incl. group 0 or null], ...]} (latin-1 strings).
Run from the repo root:  python tests/golden/gen_atomic_golden.py
"""
import json
import os
import random
import sys

import regex

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_regex_golden import Pcre  # noqa: E402

CURATED = [
    (rb'(?>a+)a', [b'aaa', b'a']),
    (rb'(?>a+)b', [b'aaab', b'b']),
    (rb'(a|ab)(?>c|cd)d', [b'abcdd', b'acd', b'abcd']),
    (rb'(\d++)(\d)', [b'123']),
    (rb'(\d++)x(\d?+)(\d*)', [b'12x34', b'1x', b'x']),
    (rb'(?>[+-]?(?:[0-9]+(?:\.[0-9]+)?|\.[0-9]+))(x?)', [b'12.5x', b'-3', b'.5', b'12.x', b'+.x']),
    (rb'((?>a|ab))+c', [b'abc', b'aac', b'ac']),
    (rb'(?>(a+))b|(a+)c', [b'aac', b'aab']),
    (rb'(x*)(?>a*)(a?)(y*)', [b'xaay', b'aaa', b'']),
    (rb'(?>(a|ab)(c|bcd))(d*)', [b'abcd', b'abcdd', b'acd']),
    (rb'((?>\d\d){1,2})-(\d+)', [b'2024-1', b'24-12', b'202-1']),
    (rb'(?>a|ab)(?>b|bc)(c?)', [b'abc', b'abbc', b'ab']),
    (rb'(?>x(?>a+)a|xa+)b', [b'xaab', b'xab']),
    (rb'(?:(?>a+)b|a)+c', [b'aabac', b'aac', b'abc']),
    (rb'(?:b|(?:[^a])++)+?', [b'bb', b'bcb', b'a']),
    # Grok building blocks (plugins/processor/grok/processor_grok_default_patterns.go:18-60), captures added
    (rb'(.*?)(?<![0-9.+-])((?>[+-]?(?:(?:[0-9]+(?:\.[0-9]+)?)|(?:\.[0-9]+))))(.*)', [b'x 12.5 y', b'a-1', b'v1.2.3', b'none']),
    # (QUOTEDSTRING writes \' and \` -- in boost's Perl syntax those are buffer-end/-start assertions, so the quotes are
    # left unescaped here; the Grok dialect difference is the expander's business, not the matcher's)
    (rb"""((?>(?<!\\)(?>"(?>\\.|[^\\"]+)+"|""|(?>'(?>\\.|[^\\']+)+')|''|(?>`(?>\\.|[^\\`]+)+`)|``))) (\w+)""",
     [b'"a b" x', b'"a\\"b" y', b'"" z', b"'q' w", b'"open x', b'`t` u']),
    (rb'((?>\d\d){1,2})/(\d\d?)', [b'2024/1', b'24/12', b'123/1']),
    (rb'(\w++)@(\w+)', [b'a@b', b'ab']),
    (rb'(\S++) (\S*+)(.*)', [b'k v rest', b'k ', b'k']),
    (rb'"((?:[^"\\]++|\\.)*+)" (\d+)', [b'"a\\"b" 12', b'"" 0', b'"a" x', b'"abc']),
]


def rx_full(p, s):
    r = regex.compile(p, regex.S | regex.M)
    m = r.fullmatch(s)
    return (None if m is None else [list(m.span(g)) for g in range(r.groups + 1)]), r.groups


def rx_search(p, s):
    r = regex.compile(p, regex.S | regex.M)
    m = r.search(s)
    return (None if m is None else [list(m.span(g)) for g in range(r.groups + 1)]), r.groups


def gen(rng, d=0):
    r = rng.random()
    if r < 0.33 or d > 2:
        return rng.choice(['a', 'b', 'c', '[ab]', '[^a]', '.', r'\d', r'\w'])
    if r < 0.48:
        return gen(rng, d + 1) + gen(rng, d + 1)
    if r < 0.60:
        return '(?:' + gen(rng, d + 1) + '|' + gen(rng, d + 1) + ')'
    if r < 0.72:
        return '(' + gen(rng, d + 1) + ')'
    if r < 0.84:
        return '(?>' + gen(rng, d + 1) + ')'
    if r < 0.88:
        return rng.choice([r'\b', '^', '$', '(?<![ab])', '(?=c)'])
    q = rng.choice(['*', '+', '?', '*+', '++', '?+', '{1,2}', '{1,2}+', '*?', '+?'])
    return '(?:' + gen(rng, d + 1) + ')' + q


def main():
    rng = random.Random(20260923)
    pcre = Pcre()
    out = {"full": [], "search": []}
    dropped = 0

    def add(kind, p, s):
        nonlocal dropped
        try:
            exp, ng = (rx_full if kind == "full" else rx_search)(p, s)
            exp_pc = (pcre.fullmatch if kind == "full" else pcre.search)(p, s, ng)
        except (regex.error, ValueError):
            return
        if exp != exp_pc:
            dropped += 1
            return
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
    n = 0
    while n < 400:
        p = gen(rng).encode()
        if b'(?>' not in p and b'+' not in p.replace(b'+?', b''):
            continue
        try:
            regex.compile(p)
        except regex.error:
            continue
        n += 1
        for _ in range(8):
            s = bytes(rng.choice(b'abc1 ') for _ in range(rng.randint(0, 7)))
            add("full", p, s)
        for _ in range(3):
            s = bytes(rng.choice(b'abc1 ') for _ in range(rng.randint(0, 9)))
            add("search", p, s)
    out = {"generator": "tests/golden/gen_atomic_golden.py", "seed": 20260923,
           "engines": ["regex %s (bytes, DOTALL|MULTILINE)" % regex.__version__, "PCRE1 8.45 (DOTALL|MULTILINE)"],
           "dropped_disagreements": dropped,
           "n_full": sum(len(c["subs"]) for c in out["full"]), "n_search": sum(len(c["subs"]) for c in out["search"]),
           "format": "full/search[i] = {p, g, subs: [[subject, flat caps incl. group 0 or null], ...]}",
           "full": out["full"], "search": out["search"]}
    with open(os.path.join(HERE, "regex_atomic_golden.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("full", out["n_full"], "search", out["n_search"], "dropped", dropped,
          "patterns", len(out["full"]), len(out["search"]))


if __name__ == "__main__":
    main()
