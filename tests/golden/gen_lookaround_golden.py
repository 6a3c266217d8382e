#!/usr/bin/env python3
"""Golden vectors for multi-byte look-arounds with a fixed-length body: (?=lit) (?!lit) (?<=lit) (?<!lit), lit a string or a sequence
of character classes -- what the Grok library's MONGO_QUERY / MONGO_SLOWQUERY use (example_config/processor_grok_patterns/mongodb:2-3:
"\\{ (?<={ ).*(?= } ntoreturn:) \\}").  Fixed-string look-arounds are regular; the device compilers turn a look-ahead into a product of
the follow NFA with the body's chain (csrc/follow_nfa.cpp applyWindows) and decide a look-behind from the fixed-width text in front of it
(csrc/regex_parse.cpp), refusing what neither covers.

The vectors are the agreement of two independent backtracking engines available here -- the `regex` module (bytes, DOTALL|MULTILINE)
and PCRE1 8.45 (the wrapper of gen_regex_golden.py) -- as in gen_atomic_golden.py; boost::regex (Perl syntax) and regexp2 give these
constructs the same meaning.

Writes tests/golden/regex_lookaround_golden.json: {"full": [...], "search": [...]}, each a list of
{p, g, subs: [[subject, flat caps incl. group 0 or null], ...]} (latin-1 strings).
Run from the repo root:  python tests/golden/gen_lookaround_golden.py
"""
import json
import os
import random
import sys

import regex

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_regex_golden import Pcre  # noqa: E402
from gen_atomic_golden import rx_full, rx_search  # noqa: E402

CURATED = [
    # the library's own (mongodb:2-3), with the groups Grok would name
    (rb'\{ (?<={ ).*(?= } ntoreturn:) \}', [b'{ x: 1 } ntoreturn:', b'{ a } ntoreturn: } ntoreturn:', b'{ x } nto', b'{  }', b'{ } ntoreturn:']),
    (rb'(\w+) (\w+)\.(\w+) (\w+): (\{ (?<={ ).*(?= } ntoreturn:) \}) (\w+):(\d+)',
     [b'query db.coll query: { a: 1 } ntoreturn:5', b'query db.coll query: { a: { b: 2 } } ntoreturn:17', b'q d.c q: { } ntoreturn:1', b'q d.c q: { a } nto:1']),
    (rb'(\w+)(?=bar)(\w+)', [b'foobar', b'foobarbar', b'bar', b'fooba']),
    (rb'foo(?=bar)', [b'foobar', b'foo', b'foobaz']),
    (rb'(\w+?)(?!bar)(\w*)', [b'foobar', b'bar', b'barbar', b'b']),
    (rb'a(?=bc)(\w+)', [b'abcd', b'abd', b'abc']),
    (rb'a(?!bc)(\w+)', [b'abcd', b'abd', b'ab']),
    (rb'x(?=ab)(?=a[bc]c)(\w+)', [b'xabc', b'xabd', b'xacc']),
    (rb'(\d+)(?!\.\d)(?:\.(\d+))?', [b'12.5', b'12.', b'12', b'1.x']),
    (rb'(?=ab)(\w)(\w*)', [b'abc', b'acb', b'a']),
    (rb'(?!ab)(\w)(\w*)', [b'abc', b'acb', b'a']),
    (rb'(?:(a+)(?!ab)|(\w))+', [b'aab', b'aaab', b'abab', b'ba']),
    (rb'(\w+)=(?="[^"]")"(.)"', [b'k="v"', b'k="vv"', b'k=v']),
    (rb'ab(?<=ab)c', [b'abc', b'abd']),
    (rb'ab(?<!ab)c', [b'abc']),
    (rb'(\d\d):(?<=\d\d:)(\d\d)', [b'12:34', b'1:34']),
    (rb'[ab]x(?<=ax)(\w*)', [b'axy', b'bxy']),
    (rb'(?:GET|PUT) (?<=[TU]T )(\S+)', [b'GET /a', b'PUT /b', b'POST /c']),
    (rb'(\w+) (?<![a-z] )(\w+)', [b'A b', b'a b']),
]

PIECES = ['a', 'b', 'c', '[ab]', r'\w', r'\d', '.', ' ']
LITS = ['ab', 'bc', 'a b', 'abc', '[ab]c', r'\d\d', 'ca', 'a[bc]a']


def gen(rng, d=0):
    r = rng.random()
    if r < 0.30 or d > 2:
        return rng.choice(PIECES)
    if r < 0.50:
        return gen(rng, d + 1) + gen(rng, d + 1)
    if r < 0.60:
        return '(?:' + gen(rng, d + 1) + '|' + gen(rng, d + 1) + ')'
    if r < 0.70:
        return '(' + gen(rng, d + 1) + ')'
    if r < 0.86:
        return rng.choice(['(?=', '(?!']) + rng.choice(LITS) + ')' + gen(rng, d + 1)
    q = rng.choice(['*', '+', '?', '{1,2}', '*?', '+?'])
    return '(?:' + gen(rng, d + 1) + ')' + q


def main():
    rng = random.Random(20260925)
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
    while n < 300:
        p = gen(rng).encode()
        if b'(?=' not in p and b'(?!' not in p:
            continue
        try:
            regex.compile(p)
        except regex.error:
            continue
        n += 1
        for _ in range(8):
            s = bytes(rng.choice(b'abc1 ') for _ in range(rng.randint(0, 7)))
            add("full", p, s)
        for _ in range(4):
            s = bytes(rng.choice(b'abc1 ') for _ in range(rng.randint(0, 9)))
            add("search", p, s)
    out = {"generator": "tests/golden/gen_lookaround_golden.py", "seed": 20260925,
           "engines": ["regex %s (bytes, DOTALL|MULTILINE)" % regex.__version__, "PCRE1 8.45 (DOTALL|MULTILINE)"],
           "dropped_disagreements": dropped,
           "n_full": sum(len(c["subs"]) for c in out["full"]), "n_search": sum(len(c["subs"]) for c in out["search"]),
           "format": "full/search[i] = {p, g, subs: [[subject, flat caps incl. group 0 or null], ...]}",
           "full": out["full"], "search": out["search"]}
    with open(os.path.join(HERE, "regex_lookaround_golden.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("full", out["n_full"], "search", out["n_search"], "dropped", dropped, "patterns", len(out["full"]), len(out["search"]))


if __name__ == "__main__":
    main()
