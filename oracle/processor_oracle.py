"""oracle/processor_oracle.py -- TEST INFRASTRUCTURE ONLY.

Pure-Python restatement (small cases) of the reference processor around the regex match:
    ProcessorParseRegexNative::Init / Process / ProcessEvent / RegexLogLineParser / AddLog
        core/plugin/processor/ProcessorParseRegexNative.cpp:29-253
    CommonParserOptions                core/plugin/processor/CommonParserOptions.cpp:28-117
    LogEvent content list semantics    core/models/LogEvent.cpp:50-106
    ProcessorInstance in/out counters  core/collection_pipeline/plugin/instance/ProcessorInstance.cpp:46-63
The regex arithmetic itself comes from oracle/bt_regex.c (OracleRegex).  Pinned against every case of the
reference's own unit test (tests/golden/reference_unittest_vectors.json, transcribed from
core/unittest/processor/ProcessorParseRegexNativeUnittest.cpp).
"""
from oracle.oracle import OracleRegex

LEGACY_RAW_LOG_KEY = "__raw_log__"       # CommonParserOptions.cpp:26
DEFAULT_CONTENT_KEY = "content"          # core/constants/Constants.cpp:25
CONTAINER_TIME_KEY, CONTAINER_SOURCE_KEY = "_time_", "_source_"  # ProcessorParseContainerLogNative.cpp:41-42


class LogEventModel:
    """contents: ordered list of [key, value, alive]  (LogEvent.h:23-24 ContentsContainer)"""

    def __init__(self, contents=()):
        self.contents = [[k, v, True] for k, v in contents]

    def _find(self, key):  # reverse linear scan over live entries (LogEvent.cpp:50-58)
        for ent in reversed(self.contents):
            if ent[2] and ent[0] == key:
                return ent
        return None

    def has(self, key):
        return self._find(key) is not None

    def get(self, key):
        e = self._find(key)
        return e[1] if e else b""

    def set_nocopy(self, key, val):  # LogEvent.cpp:83-95: overwrite in place, else append
        e = self._find(key)
        if e:
            e[0], e[1] = key, val
        else:
            self.contents.append([key, val, True])

    def delete(self, key):  # LogEvent.cpp:97-106: tombstone, order kept
        e = self._find(key)
        if e:
            e[2] = False

    def live(self):
        return [(k, v) for k, v, a in self.contents if a]

    def size(self):
        return len(self.live())


class ProcessorOracle:
    def __init__(self, config):
        # Init, ProcessorParseRegexNative.cpp:29-106
        for key in ("SourceKey", "Regex"):
            if key not in config:
                raise ValueError("mandatory param %s is missing" % key)
            if not isinstance(config[key], str):
                raise ValueError("param %s is not of type string" % key)
            if not config[key]:
                raise ValueError("mandatory string param %s is empty" % key)
        self.source_key = config["SourceKey"]
        self.regex_text = config["Regex"]
        try:
            self.regex = OracleRegex(self.regex_text)
        except ValueError:
            raise ValueError("mandatory string param Regex is not a valid regex")
        self.whole_line = self.regex_text == "(.*)"                      # :68
        if "Keys" not in config:
            raise ValueError("mandatory param Keys is missing")
        keys = config["Keys"]
        if not isinstance(keys, list) or not all(isinstance(k, str) for k in keys):
            raise ValueError("param Keys is not of type list")
        if not keys:
            raise ValueError("mandatory list param Keys is empty")
        if len(keys) == 1 and "," in keys[0]:                            # :84-86 legacy form
            keys = keys[0].split(",")
        self.keys = list(keys)
        self.source_key_overwritten = self.source_key in self.keys       # :89-94

        def opt_bool(name):
            v = config.get(name, False)
            return v if isinstance(v, bool) else False                  # wrong type: warning + default

        self.keep_fail = opt_bool("KeepingSourceWhenParseFail")
        self.keep_succeed = opt_bool("KeepingSourceWhenParseSucceed")
        self.coping_raw_log = opt_bool("CopingRawLog")
        renamed = config.get("RenamedSourceKey", "")
        self.renamed_source_key = renamed if isinstance(renamed, str) and renamed else self.source_key
        self.counters = dict(discarded=0, out_failed=0, out_key_not_found=0, out_successful=0, in_events=0, out_events=0)

    # CommonParserOptions.cpp:91-117
    def _should_add_source(self, ok):
        return (ok and self.keep_succeed) or (not ok and self.keep_fail)

    def _should_add_raw_log(self, ok):
        return (not ok) and self.keep_fail and self.coping_raw_log

    def _should_erase(self, ok, ev, file_offset_key):
        if not ok and not self.keep_fail:
            if ev.size() == 0:
                return True
            live = ev.live()
            if len(live) == 1 and file_offset_key is not None and live[0][0] == file_offset_key:
                return True
            if len(live) == 2 and ev.has(CONTAINER_TIME_KEY) and ev.has(CONTAINER_SOURCE_KEY):
                return True
        return False

    @staticmethod
    def _add_log(ev, key, val, overwritten=True):  # :176-184
        if not overwritten and ev.has(key):
            return
        ev.set_nocopy(key, val)

    def process_event(self, ev, is_log=True, file_offset_key=None):
        """-> True if the event survives (ProcessEvent :132-168)"""
        c = self.counters
        if not is_log:
            c["out_failed"] += 1
            return True
        if not ev.has(self.source_key):
            c["out_key_not_found"] += 1
            return True
        raw = ev.get(self.source_key)
        if self.whole_line:
            self._add_log(ev, self.keys[0] if self.keys else DEFAULT_CONTENT_KEY, raw)
            ok = True
        else:
            m = self.regex.fullmatch(raw) if True else None
            ok = True
            if m is None:                                        # :194-226
                c["out_failed"] += 1
                ok = False
            elif len(m) <= len(self.keys):                       # :227-244 (no counter)
                ok = False
            if ok:
                for i, key in enumerate(self.keys):              # :249-251
                    b, e = m[i + 1]
                    self._add_log(ev, key, raw[b:e] if b >= 0 else b"")
        if not ok or not self.source_key_overwritten:            # :153-155
            ev.delete(self.source_key)
        if self._should_add_source(ok):
            self._add_log(ev, self.renamed_source_key, raw, False)
        if self._should_add_raw_log(ok):
            self._add_log(ev, LEGACY_RAW_LOG_KEY, raw, False)
        if self._should_erase(ok, ev, file_offset_key):
            c["discarded"] += 1
            return False
        c["out_successful"] += 1
        return True

    def process_group(self, events, file_offset_key=None):
        """events: list of (LogEventModel | None for a non-log event).  -> surviving list (Process :108-126)"""
        self.counters["in_events"] += len(events)
        out = []
        for ev in events:
            if ev is None:
                self.process_event(None, is_log=False)
                out.append(ev)
            elif self.process_event(ev, True, file_offset_key):
                out.append(ev)
        self.counters["out_events"] += len(out)
        return out
