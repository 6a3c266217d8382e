"""oracle/multiline_oracle.py -- CPU restatement of the multiline splitter's record logic.  TEST INFRASTRUCTURE ONLY.

Follows ProcessorSplitMultilineLogStringNative::Init :36-84 (patterns compiled as written, present when not empty),
core/file_server/MultilineOptions.cpp:203-205,250-266 (IsMultiline on the stripped patterns) and
core/plugin/processor/inner/ProcessorSplitMultilineLogStringNative.cpp:126-300 (ProcessEvent), :341-380 (HandleUnmatchLogs),
:382-392 (GetNextLine).  The per-line test is boost::regex_search with match_continuous = OracleRegex.prefixmatch.
Pinned on the 47 cases of the reference's own unit test (tests/golden/multiline_vectors.json)."""
from oracle.oracle import OracleRegex


def _trimmed(pattern):
    """MultilineOptions::ParseRegex :250-266 -- only validity and IsMultiline() look at the stripped form"""
    if pattern.endswith("$"):
        pattern = pattern[:-1]
    while pattern.endswith(".*"):
        pattern = pattern[:-2]
    return pattern


def _parse(pattern):
    """the processor compiles the string as written and uses it when it is not empty (.cpp:66-76, .h:68-70); a pattern
    whose stripped form is not a valid regex is ignored with a warning (MultilineOptions.cpp:109-118)"""
    if not pattern:
        return None
    try:
        if _trimmed(pattern):
            OracleRegex(_trimmed(pattern).encode("utf-8"))
    except ValueError:
        return None
    return OracleRegex(pattern.encode("utf-8"))


class MultilineOracle:
    def __init__(self, StartPattern="", ContinuePattern="", EndPattern="", UnmatchedContentTreatment="single_line"):
        self.start, self.cont, self.end = _parse(StartPattern), _parse(ContinuePattern), _parse(EndPattern)
        if not self.start and not self.end:
            # ContinuePattern alone / nothing: the reference's ProcessEvent would index an empty regex vector (.cpp:176-184,
            # :395); the input plugin never builds the processor for such a config (InputFile.cpp:225)
            raise ValueError("not a multiline config")
        self.is_multiline = bool((self.start and _trimmed(StartPattern)) or (self.end and _trimmed(EndPattern)))
        self.discard = UnmatchedContentTreatment == "discard"

    def split(self, val: bytes):
        """-> (records [(begin, length, matched)], counters (input lines, unmatched lines, matched logs))"""
        hit = lambda rx, b, e: rx.prefixmatch(val[b:e]) is not None
        out, input_lines, unmatched, matched = [], 0, 0, 0

        def unmatch(b, e):
            nonlocal unmatched
            p = b
            while p < e:
                q = val.find(b"\n", p, e)
                q = e if q < 0 else q
                unmatched += 1
                if not self.discard:
                    out.append((p, q - p, 0))
                p = q + 1

        n = len(val)
        multi_start, partial = -1, False
        if not self.start and not self.cont and self.end:
            partial, multi_start = True, 0
        begin = 0
        while begin < n:
            q = val.find(b"\n", begin)
            ce = n if q < 0 else q
            cb = begin
            input_lines += 1
            if not partial:
                rx = self.start if self.start else self.cont
                if hit(rx, cb, ce):
                    multi_start, partial = cb, True
                elif self.end and not self.start and self.cont and hit(self.end, cb, ce):
                    out.append((cb, ce - cb, 1))
                    multi_start = ce + 1
                    matched += 1
                else:
                    unmatch(cb, ce)
            else:
                if self.cont and hit(self.cont, cb, ce):
                    begin = ce + 1
                    continue
                if self.end:
                    if self.cont:
                        if hit(self.end, cb, ce):
                            out.append((multi_start, ce - multi_start, 1))
                            matched += 1
                        else:
                            unmatch(multi_start, ce)
                        partial = False
                    elif hit(self.end, cb, ce):
                        out.append((multi_start, ce - multi_start, 1))
                        if self.start:
                            partial = False
                        else:
                            multi_start = ce + 1
                        matched += 1
                elif not self.cont:
                    if hit(self.start, cb, ce):
                        out.append((multi_start, cb - 1 - multi_start, 1))
                        multi_start = cb
                        matched += 1
                else:
                    out.append((multi_start, cb - 1 - multi_start, 1))
                    matched += 1
                    if not hit(self.start, cb, ce):
                        unmatch(cb, ce)
                        partial = False
                    else:
                        multi_start = cb
            begin = ce + 1
        if partial and multi_start < n:
            if not self.end:
                out.append((multi_start, n - multi_start, 1))
                matched += 1
            else:
                unmatch(multi_start, n)
        return out, (input_lines, unmatched, matched)
