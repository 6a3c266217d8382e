#!/usr/bin/env python3
"""Generate tests/golden/regex_golden.json -- full-match capture-offset vectors on which two independent
engines available in the build container agree: CPython `re` (bytes, DOTALL|MULTILINE == boost's mod_s/mod_m
defaults) and PCRE1 8.45 (/opt/conda/lib/libpcre.so.1, pattern wrapped as ^(?:re)\\z, DOTALL|MULTILINE).

boost::regex itself (the engine behind the reference's BoostRegexMatch, core/common/StringTools.cpp:183-211)
is not available offline, so these vectors pin the *published* Perl leftmost-first / regex_match semantics the
oracle and the HIP path must both reproduce.  Cases where the two engines disagree are dropped and counted.

Run from the repo root:  python tests/golden/gen_regex_golden.py
Deterministic (seeded); the output file is committed.
"""
import ctypes
import json
import os
import random
import re
import sys
import warnings

warnings.simplefilter('ignore', FutureWarning)

HERE = os.path.dirname(os.path.abspath(__file__))

# ----------------------------------------------------------------------------- PCRE1 via ctypes
PCRE_MULTILINE, PCRE_DOTALL, PCRE_ANCHORED = 0x2, 0x4, 0x10


class Pcre:
    def __init__(self):
        self.lib = None
        for p in ("/opt/conda/lib/libpcre.so.1", "libpcre.so.3", "libpcre.so.1"):
            try:
                self.lib = ctypes.CDLL(p)
                break
            except OSError:
                continue
        if self.lib is None:
            raise RuntimeError("PCRE1 not found")
        self.lib.pcre_compile.restype = ctypes.c_void_p
        self.lib.pcre_compile.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                          ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
        self.lib.pcre_exec.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int]

    def fullmatch(self, pattern: bytes, subject: bytes, ngroups: int):
        err = ctypes.c_char_p()
        eo = ctypes.c_int()
        code = self.lib.pcre_compile(b"(?:" + pattern + b")\\z", PCRE_DOTALL | PCRE_MULTILINE, ctypes.byref(err),
                                     ctypes.byref(eo), None)
        if not code:
            raise ValueError(err.value)
        n = 3 * (ngroups + 1)
        ov = (ctypes.c_int * n)()
        rc = self.lib.pcre_exec(code, None, subject, len(subject), 0, PCRE_ANCHORED, ov, n)
        if rc < 0:
            return None if rc == -1 else "error%d" % rc
        out = [[-1, -1] for _ in range(ngroups + 1)]
        for g in range(min(rc, ngroups + 1)):
            out[g] = [ov[2 * g], ov[2 * g + 1]]
        return out


    def search(self, pattern: bytes, subject: bytes, ngroups: int):
        err = ctypes.c_char_p()
        eo = ctypes.c_int()
        code = self.lib.pcre_compile(pattern, PCRE_DOTALL | PCRE_MULTILINE, ctypes.byref(err), ctypes.byref(eo), None)
        if not code:
            raise ValueError(err.value)
        n = 3 * (ngroups + 1)
        ov = (ctypes.c_int * n)()
        rc = self.lib.pcre_exec(code, None, subject, len(subject), 0, 0, ov, n)
        if rc < 0:
            return None if rc == -1 else "error%d" % rc
        out = [[-1, -1] for _ in range(ngroups + 1)]
        for g in range(min(rc, ngroups + 1)):
            out[g] = [ov[2 * g], ov[2 * g + 1]]
        return out


def py_search(pattern: bytes, subject: bytes):
    rx = re.compile(pattern, re.DOTALL | re.MULTILINE)
    m = rx.search(subject)
    if m is None:
        return None, rx.groups
    return [list(m.span(g)) for g in range(rx.groups + 1)], rx.groups


def py_fullmatch(pattern: bytes, subject: bytes):
    rx = re.compile(pattern, re.DOTALL | re.MULTILINE)
    m = rx.fullmatch(subject)
    if m is None:
        return None, rx.groups
    return [list(m.span(g)) for g in range(rx.groups + 1)], rx.groups


# ----------------------------------------------------------------------------- curated patterns (reference-derived)
R_A = rb'([\d\.]+) \S+ \S+ \[(\S+) \S+\] \"(\w+) ([^\\"]*)\" ([\d\.]+) (\d+) (\d+) (\d+|-) \"([^\\"]*)\" \"([^\\"]*)\"'
R_B = rb'^([^ ]*) ([^ ]*) ([^ ]*) \[([^\]]*)\] "(\S+) ([^\"]*) (\S*)" ([^ ]*) ([^ ]*) "([^\"]*)" "([^\"]*)"'
R_NGINX = rb'([\d\.:]+) - (\S+) \[(\S+) \S+\] \"(\S+) (\S+) ([^\\"]+)\" (\d+) (\d+) \"([^\\"]*)\" \"([^\\"]*)\" \"([^\\"]*)\"'

CURATED = [
    # (pattern, [subjects])  -- sources cited in tests/golden/README.md
    (R_A, [
        b'127.0.0.1 - - [07/Jul/2022:10:43:30 +0800] "POST /PutData?Category=YunOsAccountOpLog" 0.024 18204 200 37 "-" "aliyun-sdk-java"',
        b'10.0.0.1 - - [07/Jul/2022:10:43:30 +0800] "GET /a]b [x] y" 1.5 1 200 - "ref" "ua"',
        b'10.0.0.1 - - [07/Jul/2022:10:43:30 +0800] "GET /" 1.5 1 200 x "ref" "ua"',
        b'',
    ]),
    (R_B, [
        b'203.0.113.45 - - [25/Jun/2024:23:59:59 +0000] "GET /wp-admin/admin-ajax.php?action=x HTTP/1.1" 200 1847 "https://www.google.com/" "Mozilla/5.0 (Windows NT 10.0; Win64; x64)"',
        b'203.0.113.45 - - [25/Jun/2024:23:59:59 +0000] "GET /a b c HTTP/1.1" 200 1847 "-" "-"',
        b'203.0.113.45 - - [25/Jun/2024:23:59:59 +0000] "GET /x HTTP/1.1" 200 1847 "-" "-" trailing',
        b'{"time":"2024","level":"info","msg":"json line must fail"}',
    ]),
    (R_NGINX, [
        b'::1 - - [18/Jul/2022:07:28:01 +0000] "GET /hello/ilogtail HTTP/1.1" 404 153 "-" "curl/7.74.0" "-"',
    ]),
    (rb'(\w+)\t(\w+).*', [b'value1\tvalue2', b'value3\tvalue4', b'value1', b'a\tb\nc', b'\t']),
    (rb'(\d+)\s+(\d+)', [b'12  34', b'1 2', b'12', b'1\n2']),
    (rb'(.*)', [b'line1\nline2', b'', b'x']),
    (rb'(.*?) (.*)', [b'a b c', b'ab', b' ']),
    (rb'(a|ab)(c|bcd)(d*)', [b'abcd', b'abcdd', b'acd']),
    (rb'(a+)(a*)', [b'aaaa', b'a']),
    (rb'(a+?)(a*)', [b'aaaa', b'a']),
    (rb'(?:(a)|b)*', [b'ab', b'ba', b'bb', b'']),
    (rb'(a)?(b)?c', [b'c', b'ac', b'bc', b'abc']),
    (rb'(\S+)\s(\S+)\s(.*)', [b'k1 k2 rest of line', b'a b ', b'a b']),
    (rb'\[([^\]]+)\] \[(\w+)\] (.*)', [b'[2024-01-04T14:36:10] [ERROR] java.lang.Exception: x', b'[] [E] x']),
    (rb'(\d{4})-(\d{2})-(\d{2})[T ](\d{2}):(\d{2}):(\d{2})(?:\.(\d+))? (.*)', [
        b'2024-01-04 14:36:10.942 hello', b'2024-01-04T14:36:10 hello', b'2024-1-04 14:36:10 x']),
    (rb'(?i)(get|post) (\S+)', [b'GET /x', b'post /y', b'PuT /z']),
    (rb'^(\w+)$', [b'abc', b'abc\n', b'ab c']),
    (rb'(\w+)\b \b(\w+)', [b'ab cd', b'ab  cd']),
    (rb'([a-c]{2,3})([a-c]*)', [b'abcabc', b'ab', b'a']),
    (rb'([a-c]{2,3}?)([a-c]*)', [b'abcabc', b'ab']),
    (rb'(x*)(y|xz)(.*)', [b'xxxz', b'xy', b'xxz!']),
    (rb'"([^"\\]*(?:\\.[^"\\]*)*)" (\d+)', [b'"a\\"b" 12', b'"" 0', b'"a" x']),
    (rb'(\S+) (\S+) (\S+)(?: (\S+))?', [b'a b c', b'a b c d', b'a b']),
    (rb'(?<ip>\d+\.\d+\.\d+\.\d+):(?<port>\d+)'.replace(b'?<', b'?P<'), [b'10.1.2.3:80', b'10.1.2:80']),
    (rb'(a|b|c)+d', [b'abcd', b'd', b'ccd']),
    (rb'((a)|(b))+', [b'ab', b'ba', b'aab']),
    (rb'(a*)b\1'.replace(rb'\1', b'(a*)'), [b'aabaa', b'b']),
    (rb'([^,]*),([^,]*),(.*)', [b'a,b,c,d', b',,', b'a,b']),
    (rb'(\x41+)(\x{42}*)(\t?)', [b'AAB\t', b'A']),
    (rb'(.+)=(.+)', [b'a=b=c', b'=', b'a=b']),
    (rb'(.+?)=(.+)', [b'a=b=c', b'a=b']),
    (rb'(\d+)(?:ms|s|us) (\w+)', [b'12ms ok', b'3s fail', b'3m fail']),
    (rb'level=(\w+) msg="([^"]*)"(?: err="([^"]*)")?', [b'level=info msg="hi"', b'level=e msg="x" err="boom"']),
    # one-byte look-arounds, as the Grok default patterns use them (plugins/processor/grok/processor_grok_default_patterns.go:25-36)
    (rb'(?<![0-9])((?:[0-1]?[0-9]{1,2}|2[0-4][0-9]|25[0-5])[.](?:[0-1]?[0-9]{1,2}|2[0-4][0-9]|25[0-5]))(?![0-9])',
     [b'10.2', b'10.255', b'10.256', b'1.1', b'256.1']),
    (rb'(.*?)(?<![0-9.+-])([+-]?[0-9]+)(?![0-9])(.*)', [b'a 12 b', b'a+12b', b'x1.5', b'12']),
    (rb'(\w+)(?=,)(.*)', [b'ab,cd', b'ab', b'ab,']),
    (rb'(?<=\[)([^\]]*)\]|x(.*)', [b'a]', b'x[a]']),
]

# ----------------------------------------------------------------------------- random pattern generator
ALPHA = b'ab c"1\n'


class Gen:
    """Random pattern generator; every construct also carries a sampler that draws a string the construct can
    match, so about half of the emitted subjects are (near-)matches."""

    def __init__(self, rng):
        self.rng = rng

    def _set_sampler(self, pat):
        rx = re.compile(pat.encode(), re.DOTALL)
        pool = [bytes([c]) for c in ALPHA if rx.fullmatch(bytes([c]))] or [b'a']
        return lambda: self.rng.choice(pool)

    def atom(self, depth):
        r = self.rng.random()
        if r < 0.30:
            c = self.rng.choice(['a', 'b', 'c', ' ', '"', '1'])
            return c, False, (lambda c=c: c.encode())
        if r < 0.45:
            p = self.rng.choice([r'\w', r'\d', r'\s', r'\S', r'\W', '.', '[ab]', '[^a]', '[a-c]', r'[^ "]', r'[\d ]'])
            return p, False, self._set_sampler(p)
        if r < 0.75 and depth < 3:
            inner, nullable, smp = self.alt(depth + 1)
            if self.rng.random() < 0.7:
                return '(' + inner + ')', nullable, smp
            return '(?:' + inner + ')', nullable, smp
        if r < 0.80:
            return self.rng.choice(['^', '$', r'\b', r'\B', '(?=[ab])', '(?![ab])', '(?<=[a ])', '(?<![a ])', r'(?!\d)',
                                    r'(?<=\w)']), True, (lambda: b'')
        c = self.rng.choice(['a', 'b', 'c', ' '])
        return c, False, (lambda c=c: c.encode())

    def piece(self, depth):
        a, nullable, smp = self.atom(depth)
        if nullable and not a.startswith('(') or a.startswith('(?=') or a.startswith('(?!') or a.startswith('(?<'):
            return a, True, smp
        r = self.rng.random()
        if r < 0.45:
            return a, nullable, smp
        q = self.rng.choice(['*', '+', '?', '{1,2}', '{2}', '{0,2}', '{1,}', '*?', '+?', '??', '{1,2}?'])
        if nullable and q not in ('?', '??'):
            return a, nullable, smp  # never put a loop around a nullable body (engines legitimately differ there)
        qn = q[0] in '*?' or q.startswith('{0')
        lo = 0 if qn else (2 if q.startswith('{2') else 1)
        hi = {'?': 1, '{2': 2, '{1': 2, '{0': 2}.get(q[:2] if q[0] == '{' else q[0], 3)
        if q.startswith('{1,}'):
            hi = 3

        def rep(smp=smp, lo=lo, hi=hi):
            return b''.join(smp() for _ in range(self.rng.randint(lo, max(lo, hi))))
        return a + q, nullable or qn, rep

    def cat(self, depth):
        n = self.rng.randint(1, 4)
        parts, nullable, smps = [], True, []
        for _ in range(n):
            p, pn, sm = self.piece(depth)
            parts.append(p)
            smps.append(sm)
            nullable = nullable and pn
        return ''.join(parts), nullable, (lambda smps=smps: b''.join(f() for f in smps))

    def alt(self, depth):
        n = 1 if self.rng.random() < 0.7 else self.rng.randint(2, 3)
        alts, nullable, smps = [], False, []
        for _ in range(n):
            c, cn, sm = self.cat(depth)
            alts.append(c)
            smps.append(sm)
            nullable = nullable or cn
        return '|'.join(alts), nullable, (lambda smps=smps: self.rng.choice(smps)())


def rand_subject(rng):
    n = rng.randint(0, 10)
    return bytes(rng.choice(ALPHA) for _ in range(n))


def mutate(rng, s):
    if not s or rng.random() < 0.5:
        return s
    i = rng.randrange(len(s))
    r = rng.random()
    c = bytes([rng.choice(ALPHA)])
    if r < 0.4:
        return s[:i] + c + s[i + 1:]
    if r < 0.7:
        return s[:i] + c + s[i:]
    return s[:i] + s[i + 1:]


def main():
    rng = random.Random(20260921)
    pcre = Pcre()
    cases = []
    bypat = {}
    dropped = 0

    def add(pattern, subject, source):
        nonlocal dropped
        try:
            exp_py, ng = py_fullmatch(pattern, subject)
        except re.error:
            return
        try:
            exp_pc = pcre.fullmatch(pattern, subject, ng)
        except ValueError:
            return
        if exp_py != exp_pc:
            dropped += 1
            return
        key = pattern.decode('latin-1')
        if key not in bypat:
            bypat[key] = {"p": key, "g": ng, "src": source, "subs": []}
            cases.append(bypat[key])
        flat = None if exp_py is None else [v for ab in exp_py for v in ab]
        ent = [subject.decode('latin-1'), flat]
        if ent not in bypat[key]["subs"]:
            bypat[key]["subs"].append(ent)

    for pat, subs in CURATED:
        for s in subs:
            add(pat, s, "curated")
        for _ in range(6):
            add(pat, rand_subject(rng), "curated+rand")

    g = Gen(rng)
    npat = 0
    while npat < 700:
        p, _, smp = g.alt(0)
        pb = p.encode()
        try:
            re.compile(pb)
        except re.error:
            continue
        npat += 1
        for _ in range(5):
            add(pb, mutate(rng, smp()), "random")
        for _ in range(2):
            add(pb, rand_subject(rng), "random")

    total = sum(len(c["subs"]) for c in cases)
    # ---- second file: leftmost-first SEARCH vectors (Go processor_regex / Grok semantics), same two engines
    s_cases, s_bypat, s_dropped = [], {}, 0

    def add_search(pattern, subject):
        nonlocal s_dropped
        try:
            exp_py, ng = py_search(pattern, subject)
            exp_pc = pcre.search(pattern, subject, ng)
        except (re.error, ValueError):
            return
        if exp_py != exp_pc:
            s_dropped += 1
            return
        key = pattern.decode('latin-1')
        if key not in s_bypat:
            s_bypat[key] = {"p": key, "g": ng, "subs": []}
            s_cases.append(s_bypat[key])
        flat = None if exp_py is None else [v for ab in exp_py for v in ab]
        ent = [subject.decode('latin-1'), flat]
        if ent not in s_bypat[key]["subs"]:
            s_bypat[key]["subs"].append(ent)

    srng = random.Random(20260922)
    sg = Gen(srng)
    for pat, subs in CURATED:
        for sub in subs:
            add_search(pat, b"junk " + sub + b" tail")
            add_search(pat, sub)
    nsp = 0
    while nsp < 220:
        p, _, smp = sg.alt(0)
        pb = p.encode()
        try:
            re.compile(pb)
        except re.error:
            continue
        nsp += 1
        for _ in range(4):
            add_search(pb, rand_subject(srng) + mutate(srng, smp()) + rand_subject(srng))
        add_search(pb, rand_subject(srng))
    s_total = sum(len(c["subs"]) for c in s_cases)
    with open(os.path.join(HERE, "regex_search_golden.json"), "w") as f:
        json.dump({"generator": "tests/golden/gen_regex_golden.py", "seed": 20260922, "semantics": "leftmost-first search",
                   "dropped_disagreements": s_dropped, "n_patterns": len(s_cases), "n_cases": s_total,
                   "format": "cases[i] = {p, g, subs: [[subject, flat caps incl. group 0 or null], ...]}",
                   "cases": s_cases}, f, separators=(",", ":"))
    print("search: patterns", len(s_cases), "cases", s_total, "dropped", s_dropped)

    matched = sum(1 for c in cases for e in c["subs"] if e[1] is not None)
    out = {"generator": "tests/golden/gen_regex_golden.py", "seed": 20260921,
           "engines": ["CPython re %s (bytes, DOTALL|MULTILINE)" % sys.version.split()[0], "PCRE1 8.45 (^(?:re)\\z, DOTALL|MULTILINE)"],
           "dropped_disagreements": dropped, "n_patterns": len(cases), "n_cases": total, "n_matched": matched,
           "format": "cases[i] = {p: pattern (latin-1), g: group count, subs: [[subject (latin-1), flat caps [b0,e0,b1,e1,...] or null], ...]}",
           "cases": cases}
    with open(os.path.join(HERE, "regex_golden.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("patterns", len(cases), "cases", total, "matched", matched, "dropped", dropped)


if __name__ == "__main__":
    main()
