"""The reference's benchmark pipeline in one device trip (include/lc_processor.h lc_pipeline_*):
split (ProcessorSplitLogStringNative.cpp:101-174) -> processor_parse_regex_native (ProcessorParseRegexNative.cpp:108-253) ->
processor_filter_regex_native on keys the parser produced (ProcessorFilterNative.cpp:159-286), configured as in
test/benchmark/local/test_cases/performance_file_to_blackhole_loongcollector/loongcollector.yaml.

Three ways to the same events: the fused trip, the same three classes one after the other ("Fused": false), and the chain of
the CPU oracles (split_oracle -> processor_oracle -> filter_oracle)."""
import itertools

import numpy as np
import pytest

from loongcollector_amd import binding as B
from loongcollector_amd import corpus
from loongcollector_amd.processor import EventGroup, Pipeline
from oracle.filter_oracle import FilterOracle
from oracle.processor_oracle import LogEventModel, ProcessorOracle
from oracle.split_oracle import split_lines

pytestmark = pytest.mark.gpu


def _buffer(n_lines, seed=5, trailing_newline=True):
    """n access-log lines for regex B; every 7th has the user agent the benchmark's filter keeps, every 11th is junk the parser
    cannot match, every 13th is empty"""
    rng = np.random.default_rng(seed)
    data, off, length = corpus.apache_batch(n_lines, "B", line_bytes=200, seed=seed, pool_lines=min(n_lines, 512))
    lines = [bytes(data[o:o + l]) for o, l in zip(off[:-1], length)]
    out = []
    for i, l in enumerate(lines):
        if i % 13 == 12:
            out.append(b"")
        elif i % 11 == 10:
            out.append(b"{\"level\": \"info\", \"msg\": \"not an access log %d\"}" % i)
        elif i % 7 == 6:
            head, _, _ = l.rpartition(b' "')
            out.append(head + b' "no-agent"')
        else:
            out.append(l)
    buf = b"\n".join(out)
    return buf + (b"\n" if trailing_newline else b""), int(rng.integers(0, 1 << 40))


def _oracle_chain(buf, parse_cfg, filter_cfg, file_offset, file_offset_key):
    events = []
    for b, l in split_lines(buf):
        contents = [("content", buf[b:b + l])]
        if file_offset_key:
            contents.append((file_offset_key, str(file_offset + b).encode()))
        events.append(LogEventModel(contents))
    po = ProcessorOracle(parse_cfg)
    kept = po.process_group(events, file_offset_key=file_offset_key)
    fo = FilterOracle(filter_cfg)
    out = fo.process([dict(ev.live()) for ev in kept])
    return [list(c.items()) for c in out], po.counters


PARSE = {"SourceKey": "content", "Regex": corpus.REGEX_B, "Keys": corpus.KEYS_B}
FILTERS = [
    {"FilterKey": ["user_agent"], "FilterRegex": ["^no-agent$"]},                       # the reference benchmark's filter
    {"FilterKey": ["method", "response_code"], "FilterRegex": ["GET|POST", r"2\d\d"]},  # two rules
    {"FilterKey": ["referrer"], "FilterRegex": [".*"]},                                 # keeps every parsed line
]


@pytest.mark.parametrize("filt", range(len(FILTERS)))
def test_fused_trip_equals_the_three_processors_and_the_oracles(filt):
    assert B.device_count() >= 1
    filter_cfg = FILTERS[filt]
    for keep_fail, keep_ok, renamed, offset_key, trailing in itertools.product(
            [False, True], [False, True], ["", "raw"], [None, "__file_offset__"], [True, False]):
        parse_cfg = dict(PARSE, KeepingSourceWhenParseFail=keep_fail, KeepingSourceWhenParseSucceed=keep_ok, RenamedSourceKey=renamed)
        cfg = {"Parse": parse_cfg, "Filter": filter_cfg}
        fused, chained = Pipeline(cfg), Pipeline(dict(cfg, Fused=False))
        assert fused.fused and not chained.fused
        buf, pos = _buffer(700, seed=5 + filt, trailing_newline=trailing)
        g1 = fused.process(EventGroup.from_buffer(buf, file_offset=pos, file_offset_key=offset_key))
        g2 = chained.process(EventGroup.from_buffer(buf, file_offset=pos, file_offset_key=offset_key))
        d1, d2 = g1.to_dict(), g2.to_dict()
        assert d1 == d2, (filter_cfg, parse_cfg, offset_key)
        c1, c2 = fused.counters(), chained.counters()
        assert c1["groups_fused"] == 1 and c2["groups_chained"] == 1
        for k in ("discarded_events_total", "out_failed_events_total", "out_key_not_found_events_total", "out_successful_events_total",
                  "filter_in_events", "filter_out_events"):
            assert c1[k] == c2[k], (k, c1, c2)
        want, oc = _oracle_chain(buf, parse_cfg, filter_cfg, pos, offset_key)
        got = [[(k, v.encode("latin-1") if isinstance(v, str) else v) for k, v in ev] for ev in g1.contents()]
        assert [[(k, bytes(v)) for k, v in ev] for ev in want] == got
        assert c1["out_failed_events_total"] == oc["out_failed"] and c1["discarded_events_total"] == oc["discarded"]
        assert c1["out_successful_events_total"] == oc["out_successful"]
        assert len(got) > 20


def test_what_cannot_be_fused_runs_the_three_steps():
    # a rule on a key the parser does not write, an expression filter, a rule on the kept source
    for filter_cfg in ({"FilterKey": ["nope"], "FilterRegex": [".*"]},
                       {"ConditionExp": {"key": "method", "exp": "GET", "type": "regex"}},
                       {"FilterKey": ["content"], "FilterRegex": [".*"]}):
        p = Pipeline({"Parse": dict(PARSE, KeepingSourceWhenParseFail=True), "Filter": filter_cfg})
        assert not p.fused
        buf, pos = _buffer(300)
        g = p.process(EventGroup.from_buffer(buf, file_offset=pos))
        want, _ = _oracle_chain(buf, dict(PARSE, KeepingSourceWhenParseFail=True), filter_cfg, pos, None)
        got = [[(k, v.encode("latin-1")) for k, v in ev] for ev in g.contents()]
        assert [[(k, bytes(v)) for k, v in ev] for ev in want] == got
        assert p.counters()["groups_chained"] == 1


def test_read_buffer_sizes_and_edge_buffers():
    """512 KB read buffers (what LogFileReader hands over), an empty buffer, a buffer of separators, one unterminated line"""
    cfg = {"Parse": PARSE, "Filter": FILTERS[0]}
    fused, chained = Pipeline(cfg), Pipeline(dict(cfg, Fused=False))
    big, pos = _buffer(2600)     # ~512 KB
    assert 400_000 < len(big) < 700_000
    for buf in (big, b"", b"\n\n\n", big.split(b"\n")[6], b"\n" + big.split(b"\n")[6] + b"\n\n"):
        g1 = fused.process(EventGroup.from_buffer(buf, file_offset=pos))
        g2 = chained.process(EventGroup.from_buffer(buf, file_offset=pos))
        assert g1.to_dict() == g2.to_dict()
    assert fused.counters()["survivors"] == chained.counters()["filter_out_events"] > 300
    assert fused.counters()["lines"] >= 2600
