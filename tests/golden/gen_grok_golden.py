#!/usr/bin/env python3
"""Golden vectors for the Grok processor.

Two parts, written to tests/golden/grok_golden.json:
  "reference": the parse vectors the reference's own unit test asserts (plugins/processor/grok/processor_grok_test.go
               :119-373), transcribed as data: config, input logs, expected output contents.
  "regex":     fields computed by an engine independent of this repo -- the Python `regex` module -- restating
               processGrok (processor_grok.go:148-194: ordered Match list, search, iterate all matches, named non-empty
               groups) over the expanded patterns, on hand-made and mutated lines.  `regex` stands in for regexp2 (both
               are backtracking engines with atomic groups and look-behind; SURVEY.md section 8c names it the stand-in).
The expansion used for the second part is oracle/grok_oracle.py's, which tests/test_grok_host.py pins on the reference's
expected strings.  Run from the repo root:  python tests/golden/gen_grok_golden.py
"""
import json
import os
import random
import sys

import regex

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.grok_oracle import GrokOracle  # noqa: E402

HTTP = {"HTTP": "%{IP:client} %{WORD:method} %{URIPATHPARAM:request} %{NUMBER:bytes} %{NUMBER:duration}"}
THREE = ["%{HTTP}", "%{WORD:word1} %{NUMBER:request_time} %{WORD:word2}",
         "%{YEAR:year} %{MONTH:month} %{MONTHDAY:day} %{QUOTEDSTRING:motto}"]
STACK = (r'\[%{TIMESTAMP_ISO8601:time_local}\] %{NUMBER:pid} %{QUOTEDSTRING:thread} prio=%{NUMBER:prio} '
         r'tid=%{BASE16NUM:tid} nid=%{BASE16NUM:nid} %{DATA:func} \[%{BASE16NUM:addr}\]%{SPACE}(?ms)%{GREEDYDATA:stack}')
STACK_LOG = ('[2023-02-09T00:24:43.922554223+08:00] 1 "BLOCKED_TEST pool-1-thread-2" prio=6 tid=0x0000000007673800 '
             'nid=0x260c waiting for monitor entry [0x0000000008abf000]\n'
             'java.lang.Thread.State: BLOCKED (on object monitor)\n'
             '\t\t\t at com.nbp.theplatform.threaddump.ThreadBlockedState.monitorLock(ThreadBlockedState.java:43)\n'
             '\t\t\t - waiting to lock <0x0000000780a000b0> (a com.nbp.theplatform.threaddump.ThreadBlockedState)')
FOUR_LOGS = ["begin 123.456 end", '2019 June 24 "I am iron man"', "WRONG LOG", "10.0.0.0 GET /index.html 15824 0.043"]
FOUR_FIELDS = [
    [["word1", "begin"], ["request_time", "123.456"], ["word2", "end"]],
    [["year", "2019"], ["month", "June"], ["day", "24"], ["motto", '"I am iron man"']],
    [],
    [["client", "10.0.0.0"], ["method", "GET"], ["request", "/index.html"], ["bytes", "15824"], ["duration", "0.043"]],
]


def reference_vectors():
    c = lambda s: [["content", s]]
    one = {"Match": ["%{WORD:word1} %{NUMBER:request_time} %{WORD:word2}"]}
    out = [
        # :130-165
        {"cite": ":130-142", "config": one, "in": [c("begin 123.456 end")],
         "out": [c("begin 123.456 end") + [["word1", "begin"], ["request_time", "123.456"], ["word2", "end"]]]},
        {"cite": ":144-152", "config": one, "in": [c("")], "out": [c("")]},
        {"cite": ":154-165", "config": one, "in": [c("begin 123.456 end\n")],
         "out": [c("begin 123.456 end\n") + [["word1", "begin"], ["request_time", "123.456"], ["word2", "end"]]]},
        # :167-181 (UTF-8 value; key keeps the original alias spelling)
        {"cite": ":167-181", "config": {"Match": ["%{WORD:english-word} %{GREEDYDATA:message}"]},
         "in": [c("hello こんにちは")], "out": [c("hello こんにちは") + [["english-word", "hello"], ["message", "こんにちは"]]]},
        {"cite": ":183-198", "config": {"Match": ["%{WORD:english-word} %{GREEDYDATA:message} (?P<message2>.*)"]},
         "in": [c("hello こんにちは 你好")],
         "out": [c("hello こんにちは 你好") + [["english-word", "hello"], ["message", "こんにちは"], ["message2", "你好"]]]},
        # :200-228 ((?ms) in the middle of the pattern)
        {"cite": ":200-228", "config": {"Match": [STACK]}, "in": [c(STACK_LOG)],
         "out": [c(STACK_LOG) + [["time_local", "2023-02-09T00:24:43.922554223+08:00"], ["pid", "1"],
                                 ["thread", '"BLOCKED_TEST pool-1-thread-2"'], ["prio", "6"],
                                 ["tid", "0x0000000007673800"], ["nid", "0x260c"], ["func", "waiting for monitor entry"],
                                 ["addr", "0x0000000008abf000"], ["stack", STACK_LOG.split("\n", 1)[1]]]]},
        # :230-296 three patterns, first match wins
        {"cite": ":230-296", "config": {"CustomPatterns": HTTP, "Match": THREE}, "in": [c(s) for s in FOUR_LOGS],
         "out": [c(s) + f for s, f in zip(FOUR_LOGS, FOUR_FIELDS)]},
        # :298-320 IgnoreParseFailure=false drops the source of the unmatched log
        {"cite": ":298-320", "config": {"CustomPatterns": HTTP, "Match": THREE, "IgnoreParseFailure": False},
         "in": [c(s) for s in FOUR_LOGS],
         "out": [(c(s) if f else []) + f for s, f in zip(FOUR_LOGS, FOUR_FIELDS)]},
        # :322-372 KeepSource=false drops the source of matched logs
        {"cite": ":322-372", "config": {"CustomPatterns": HTTP, "Match": THREE, "KeepSource": False},
         "in": [c(s) for s in FOUR_LOGS],
         "out": [(f if f else c(s)) for s, f in zip(FOUR_LOGS, FOUR_FIELDS)]},
        # :408-427 zero-width pattern: Init succeeds, processing goes through
        {"cite": ":408-427", "config": {"Match": ["(?s)^$"]}, "in": [c("")], "out": [c("")]},
    ]
    return out


CASES = [
    ({"Match": ["%{IPV4:ip}"]}, ["10.0.0.1", "a 10.0.0.1 b 192.168.1.254 c", "1.2.3.4.5.6.7.8", "999.1.1.1 1.1.1.1", "no ip", "",
                                  "x10.0.0.1", "10.0.0.1x 20.0.0.2"]),
    ({"Match": ["%{WORD:w}"]}, ["alpha beta  gamma", "  ", "a", "one,two;three", "ünïcode wörd"]),
    ({"Match": ["%{NUMBER:n}"]}, ["1 2.5 -3 +4.0 .5 1.", "v1.2.3", "a-1b", "12abc34"]),
    ({"Match": ["%{INT:a}-%{INT:b}", "%{WORD:only}"]}, ["1-2 3-4", "x 5-6", "--", "7-", "w"]),
    ({"Match": ["(?P<k>\\w+)=(?P<v>\\S*)"]}, ["a=1 b= c=3", "=x", "k=v", "a==b"]),
    ({"Match": ["(?P<x>a)|(?P<x>b)c"]}, ["a", "bc", "xbca", "b"]),                       # same-named groups
    ({"Match": ["(?P<e>x*)y"]}, ["y", "xxy y xy", "zzz"]),                               # empty captures are skipped
    ({"Match": ["\\b(?P<t>\\d\\d:\\d\\d)\\b", "%{GREEDYDATA:rest}"]}, ["at 10:30 and 11:45", "110:30", "nothing"]),
    ({"Match": ["%{SYSLOGBASE} %{GREEDYDATA:msg}"]}, ["Mar 16 00:01:25 evita postfix/smtpd[1713]: connect from camomile.cloud9.net[168.100.1.3]",
                                                     "Mar 16 00:01:25 evita postfix: x", "garbage line"]),
    ({"Match": ["%{COMMONAPACHELOG}"]}, ['127.0.0.1 - frank [10/Oct/2000:13:55:36 -0700] "GET /apache_pb.gif HTTP/1.0" 200 2326',
                                         '::1 - - [10/Oct/2000:13:55:36 -0700] "-" 408 -', 'bad']),
    ({"Match": ["%{TIMESTAMP_ISO8601:ts} %{LOGLEVEL:level} %{GREEDYDATA:msg}"]},
     ["2024-01-04T14:36:10.942Z ERROR boom", "2024-01-04 14:36:10,942 info ok", "2024-13-04 14:36:10 WARN bad month", "x"]),
    ({"Match": ["%{QS:q}"]}, ['say "hi" and "bye"', '"a\\"b"', "'single' `tick`", '"" x', 'broken "']),
    ({"Match": ["%{URI:u}"]}, ["see http://user:pw@example.com:8080/p/a?x=1&y=2 ok", "ftp://h/", "nope"]),
    ({"Match": ["%{MAC:m}"]}, ["aa:bb:cc:dd:ee:ff AABB.CCDD.EEFF 00-11-22-33-44-55", "zz"]),
    ({"Match": ["%{HTTPDATE:d}"]}, ["[10/Oct/2000:13:55:36 -0700]", "10/Xxx/2000:13:55:36 -0700"]),
    ({"Match": ["%{UUID:id}"]}, ["id=123e4567-e89b-12d3-a456-426614174000;", "123e4567-e89b-12d3-a456-42661417400"]),
    ({"Match": ["%{PATH:p}"]}, ["/var/log/x.log C:\\Windows\\a.txt", "rel/path"]),
]


def regex_process(g, patterns, val):
    """processGrok with the `regex` module as the engine"""
    for rx in patterns:
        out = []
        names = sorted(rx.groupindex, key=lambda k: rx.groupindex[k])
        for m in rx.finditer(val):
            for nm in names:
                if nm.isdigit():
                    continue
                v = m.group(nm)
                if v:
                    out.append([g.aliases.get(nm, nm), v.decode("latin-1")])
        if out:
            return out
    return []


def main():
    rng = random.Random(20260924)
    ref = reference_vectors()
    cases = []
    for cfg, lines in CASES:
        g = GrokOracle(cfg["Match"], custom_patterns=cfg.get("CustomPatterns"))
        patterns = [regex.compile(e.encode("utf-8")) for e in g.expanded]
        subs = []
        pool = [s.encode("utf-8") for s in lines]
        extra = []
        for s in pool:
            for _ in range(3):  # mutations: drop / duplicate / replace a byte, glue two lines
                b = bytearray(s)
                if b:
                    k = rng.randrange(len(b))
                    op = rng.randrange(3)
                    if op == 0:
                        del b[k]
                    elif op == 1:
                        b.insert(k, b[k])
                    else:
                        b[k] = rng.choice(b" .:-/\"0a")
                extra.append(bytes(b))
            extra.append(s + b" " + rng.choice(pool))
        for s in pool + extra:
            subs.append([s.decode("latin-1"), regex_process(g, patterns, s)])
        cases.append({"config": cfg, "subs": subs})
    out = {"generator": "tests/golden/gen_grok_golden.py", "engine": "regex %s" % regex.__version__,
           "format": "reference: {config, in, out} logs as [[key, value], ...]; regex: {config, subs: [[value (latin-1), "
                     "[[key, value (latin-1)], ...]], ...]}",
           "reference": ref, "regex": cases}
    with open(os.path.join(HERE, "grok_golden.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print("reference vectors", len(ref), "regex cases", len(cases), "values", sum(len(c["subs"]) for c in cases))


if __name__ == "__main__":
    main()
