"""oracle/go_regex_oracle.py -- CPU restatement of the Go plugin processor_regex.  TEST INFRASTRUCTURE ONLY.

Follows plugins/processor/regex/regex.go: Init :50-66, ProcessLog :80-101, shouldKeepSource :103-105, processRegex
:107-129.  Go's regexp (RE2 syntax, leftmost-first, "(?s)" prepended) is restated by oracle/bt_regex.c with flags
ORX_NO_MOD_M | ORX_REGEXP2 ('.' matches '\\n', single-line anchors, RE2's \\s).  Pinned on the reference's own test
vectors (plugins/processor/regex/regex_test.go:60-200, transcribed in tests/golden/go_regex_vectors.json) and, for the
search semantics, on tests/golden/regex_search_golden.json (CPython re ∧ PCRE1)."""
from oracle.oracle import ORX_NO_MOD_M, ORX_REGEXP2, OracleRegex


class GoRegexOracle:
    def __init__(self, Regex="", Keys=(), FullMatch=False, NoKeyError=False, NoMatchError=True, KeepSource=False,
                 KeepSourceIfParseError=True, SourceKey=""):
        if not Keys:
            raise ValueError("no regex key error")
        self.keys = list(Keys)
        self.full = FullMatch
        self.keep = KeepSource
        self.keep_on_error = KeepSourceIfParseError
        self.source_key = SourceKey
        self.re = OracleRegex(Regex.encode("utf-8"), ORX_NO_MOD_M | ORX_REGEXP2)

    def process_regex(self, val):
        caps = self.re.search(val)
        if caps is None or (self.full and caps[0] != (0, len(val))):
            return False, []
        if len(caps) - 1 < len(self.keys):
            return False, []
        return True, [(k, val[b:e]) for k, (b, e) in zip(self.keys, caps[1:]) if b >= 0 and e >= b]

    def process_log(self, contents):
        contents = list(contents)
        for i, (key, value) in enumerate(contents):
            if not self.source_key or self.source_key == key:
                ok, fields = self.process_regex(value)
                contents.extend(fields)
                if not (self.keep or (self.keep_on_error and not ok)):
                    del contents[i]
                break
        return contents
