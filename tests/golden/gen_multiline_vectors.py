#!/usr/bin/env python3
"""Multiline-split vectors, extracted as DATA from the reference's own unit test
(core/unittest/processor/ProcessorSplitMultilineLogStringNativeUnittest.cpp).  Needs /root/reference.

Every `// case:` block of that file builds one input event whose content is a sequence of the five canned lines
(begin / continue / end / unmatch) and asserts the contents of the output events; this script reads those sequences
(it does not run or copy any code) and writes tests/golden/multiline_vectors.json:
  {lines: {token: text}, patterns: {...}, cases: [{cite, config, in: [tokens], out: [[tokens of one output event], ...]}]}
Run from the repo root:  python tests/golden/gen_multiline_vectors.py
"""
import json
import os
import re

SRC = "/root/reference/core/unittest/processor/ProcessorSplitMultilineLogStringNativeUnittest.cpp"
# the same `// case:` blocks exist for the merge processor (input split into one event per line by ProcessorSplitLogStringNative,
# then ProcessorMergeMultilineLogNative with MergeType "regex"): written to multiline_merge_vectors.json
SRC_MERGE = "/root/reference/core/unittest/processor/ProcessorMergeMultilineLogNativeUnittest.cpp"
HERE = os.path.dirname(os.path.abspath(__file__))
TOK = {"LOG_BEGIN_STRING": "B", "LOG_CONTINUE_STRING": "C", "LOG_END_STRING": "E", "LOG_UNMATCH": "U"}


def main(src=SRC, out_name="multiline_vectors.json", kind_key="SplitType"):
    text = open(src, encoding="utf-8").read()
    consts = {}
    for m in re.finditer(r'const std::string (\w+) = (?:R"\((.*?)\)"|"(.*?)");', text):
        consts[m.group(1)] = m.group(2) if m.group(2) is not None else m.group(3)
    lines = text.split("\n")
    funcs = [(i, l) for i, l in enumerate(lines) if l.startswith("void ") and "::Test" in l]
    cases = []
    for fi, (start, head) in enumerate(funcs):
        end = funcs[fi + 1][0] if fi + 1 < len(funcs) else len(lines)
        body = lines[start:end]
        cfg = {}
        for l in body:
            m = re.match(r'\s*config\["(\w+)"\] = (\w+|"[^"]*"|true|false);', l)
            if m:
                v = m.group(2)
                cfg[m.group(1)] = consts[v] if v in consts else (v == "true" if v in ("true", "false") else v.strip('"'))
        if cfg.get("EnableRawContent") or kind_key not in cfg or cfg[kind_key] != "regex":
            continue
        config = {k: cfg[k] for k in ("StartPattern", "ContinuePattern", "EndPattern", "UnmatchedContentTreatment") if k in cfg}
        # case blocks
        idx = [i for i, l in enumerate(body) if l.strip().startswith("// case:")]
        for ci, at in enumerate(idx):
            stop = idx[ci + 1] if ci + 1 < len(idx) else len(body)
            block = "\n".join(body[at:stop])
            if block.count("FromJsonString") != 1:   # a block that holds several unlabelled sub-cases: not transcribed
                continue
            parts = block.split("expectJson")
            if "inJson" not in parts[0]:
                continue
            if len(parts) < 2:   # no expected events at all: the test asserts ToJsonString() == "null"
                if 'STREQ("null"' not in block.replace("_FATAL", ""):
                    continue
                parts = [parts[0], ""]

            def contents(s):
                out = []
                for m in re.finditer(r'"content" : "\)"(.*?)<< R"\("', s, re.S):
                    toks = [TOK[t] for t in re.findall(r"LOG_\w+", m.group(1)) if t in TOK]
                    out.append(toks)
                return out

            ins = contents(parts[0])
            outs = contents("expectJson".join(parts[1:]))
            if len(ins) != 1:
                continue
            null_out = 'APSARA_TEST_STREQ("null"' in block or "APSARA_TEST_STREQ_FATAL(\"null\"" in block
            cases.append({"cite": "%s :%d (%s)" % (head.split("::")[0].split()[-1] + "::" + head.split("::")[1].split("(")[0],
                                                  start + at + 1, body[at].strip()[3:]),
                          "config": config, "in": ins[0], "out": [] if null_out else outs})
    out = {"source": src.replace("/root/reference/", ""),
           "lines": {TOK[k]: consts[k] for k in TOK},
           "patterns": {k: consts[k] for k in ("LOG_BEGIN_REGEX", "LOG_CONTINUE_REGEX", "LOG_END_REGEX")},
           "cases": cases}
    with open(os.path.join(HERE, out_name), "w", encoding="utf-8") as f:
        json.dump(out, f, indent=0)
    print("cases", len(cases))
    for c in cases[:6]:
        print(c["cite"], c["config"].get("UnmatchedContentTreatment"), c["in"], "->", c["out"])


if __name__ == "__main__":
    main()
    main(SRC_MERGE, "multiline_merge_vectors.json", "MergeType")
