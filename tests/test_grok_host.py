"""Grok without a GPU: the pattern-library expander (C++, through the C ABI) and the Grok oracle
against the strings and vectors the reference's own tests pin, the oracle against an independent engine, and the device
algorithm (ordered patterns, resumed searches, named non-empty groups) emulated on the compiled tables.

Reference: plugins/processor/grok/processor_grok.go + processor_grok_test.go (cited per test)."""
import json
import random
import os

import numpy as np
import pytest

from loongcollector_amd import binding as B
from loongcollector_amd.grok import Grok, GrokInitError
from oracle.grok_oracle import MATCH_SUCCESS, GrokOracle
from tests.helpers.nfa_atomic_interp import AtomicNfaInterp
from tests.helpers.table_interp import TdfaInterp

GROK_SYNTAX = (B.LC_SYNTAX_SEARCH | B.LC_SYNTAX_NAMED_ONLY | B.LC_SYNTAX_NO_DOTALL | B.LC_SYNTAX_NO_MULTILINE
               | B.LC_SYNTAX_REGEXP2)


@pytest.fixture(scope="module")
def expansions(golden_dir):
    with open(os.path.join(golden_dir, "grok_expansions.json"), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "grok_golden.json"), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="module")
def pattern_dir(expansions, tmp_path_factory):
    """CustomPatternDir ./test_patterns of the reference test, rebuilt from the fixture as two files"""
    d = tmp_path_factory.mktemp("test_patterns")
    items = sorted(expansions["test_patterns"].items())
    half = len(items) // 2
    for name, part in (("aws", items[:half]), ("grok-patterns", items[half:])):
        with open(d / name, "w", encoding="utf-8") as f:
            f.write('""" a comment line, skipped because it starts with a quote (processor_grok.go:219)\n\n')
            for k, v in part:
                f.write(k + " " + v + "\n")
    return str(d)


def test_expander_reproduces_the_strings_the_reference_test_pins(expansions, pattern_dir):
    """processor_grok_test.go:36-41 (expected strings), :74-116 (where they come from)"""
    for c in expansions["cases"]:
        cfg = {"CustomPatterns": c["custom"]}
        if c["custom_dir"]:
            cfg["CustomPatternDir"] = [pattern_dir]
        g = Grok(**cfg)   # library only: whether a device engine can RUN a Match entry is a separate question
        got = g.denormalize(c["match"]) if "match" in c else g.processed(c["processed"])
        custom = dict(expansions["test_patterns"]) if c["custom_dir"] else {}
        custom.update(c["custom"])
        o = GrokOracle([c["match"]] if "match" in c else [], custom_patterns=custom)
        want = c["expanded"]
        assert (o.expanded[0] if "match" in c else o.processed[c["processed"]]) == want
        assert got == want


def test_library_only_handles_expand_every_default_pattern_like_the_oracle():
    g, o = Grok(), GrokOracle([])
    assert len(o.processed) >= 78
    for name, text in o.processed.items():
        assert g.processed(name) == text, name
    assert g.processed("NO_SUCH_PATTERN") is None and g.n_match == 0


@pytest.mark.parametrize("cfg, needle", [
    ({"CustomPatternDir": ["./no_exist_path"]}, "invalid path"),                                    # :380-384
    ({"CustomPatterns": {"TEST": "%{IP:client} ("}, "Match": ["%{TEST}"]}, "Match[0]"),              # :386-392
    ({"CustomPatterns": {"A": "%{B:b}", "B": "%{A:a}"}}, "cyclic"),                                  # :394-402
    ({"Match": ["%{NOPE:x}"]}, "no pattern found for NOPE"),                                         # processor_grok.go:296
    ({"CustomPatterns": {"X": "%{WORD:a:bogus}"}}, "invalid pattern"),                               # :248
    ({"Match": [r"(\w+) \1"]}, "back-references"),                                                   # no device engine
    ({"CustomPatterns": dict([("P0", "a")] + [("P%d" % i, "%%{P%d}%%{P%d}" % (i - 1, i - 1)) for i in range(1, 40)]),
      "Match": ["%{P39}"]}, "grows beyond"),                                                        # 2^39 copies of "a"
])
def test_init_errors(cfg, needle):
    with pytest.raises(GrokInitError) as e:
        Grok(**cfg)
    assert needle in str(e.value)
    if "Match" not in cfg or "Match[0]" not in needle:
        return


def test_keys_aliases_columns_and_defaults():
    g = Grok(Match=["%{WORD:english-word} %{GREEDYDATA:message} (?P<message2>.*)", "%{IPV4:ip}(?P<x>a)|(?P<x>b)"])
    assert "(?P<english_word>" in g.expanded(0)                       # aliasizePatternName :319-323
    assert g.keys == ["english-word", "message", "message2", "ip", "x"]   # nameToAlias :326-332
    assert g.columns(0) == ["english-word", "message", "message2"]
    assert g.columns(1) == ["ip", "x", "x"]                           # same-named groups: two columns, one field
    assert g.row_ints == 2 * (1 + 3)
    assert Grok(Match=["(?s)^$"]).n_match == 1                        # zero-width Match only warns (:348-351, test :408-427)


def test_oracle_agrees_with_the_regex_module_and_the_reference_vectors(golden):
    n = 0
    for c in golden["regex"]:
        o = GrokOracle(c["config"]["Match"], custom_patterns=c["config"].get("CustomPatterns"))
        for val, want in c["subs"]:
            n += 1
            _, fields = o.process_value(val.encode("latin-1"))
            assert [[k, v.decode("latin-1")] for k, v in fields] == want, (c["config"]["Match"], val)
    assert n >= 300
    for r in golden["reference"]:
        cfg = r["config"]
        o = GrokOracle(cfg["Match"], custom_patterns=cfg.get("CustomPatterns"),
                       ignore_parse_failure=cfg.get("IgnoreParseFailure", True), keep_source=cfg.get("KeepSource", True))
        for log, want in zip(r["in"], r["out"]):
            got = o.process_log([(k, v.encode("utf-8")) for k, v in log])
            assert [[k, v.decode("utf-8")] for k, v in got] == want, r["cite"]


class TableGrok:
    """What lcGrokMatchDevice does, replayed on the compiled tables: per pattern a resumed-search loop, named non-empty
    groups, merged same-named columns; first pattern that collected something wins."""

    def __init__(self, g):
        self.g = g
        self.interps = []
        for i in range(g.n_match):
            rx = B.GpuRegex(g.expanded(i).encode("utf-8"), syntax_flags=GROK_SYNTAX)
            it = TdfaInterp(rx) if rx.info()["engine"] == B.LC_ENGINE_TDFA else AtomicNfaInterp(rx)
            assert (rx.info()["engine"] == B.LC_ENGINE_TDFA) == (g.engine(i) == B.LC_ENGINE_TDFA)
            self.interps.append(it)

    def process_value(self, val):
        for p, it in enumerate(self.interps):
            cols = self.g.columns(p)
            out, start = [], 0
            while True:
                caps = it.fullmatch(val, start=start)
                if caps is None:
                    break
                merged, order = {}, []
                for c, key in enumerate(cols):
                    b, e = caps[2 + 2 * c], caps[3 + 2 * c]
                    if key is None:
                        continue
                    if key not in merged:
                        merged[key] = (-1, -1)
                        order.append(key)
                    if b >= 0 and b >= merged[key][0]:
                        merged[key] = (b, e)
                out += [[k, val[merged[k][0]:merged[k][1]]] for k in order if merged[k][1] > merged[k][0]]
                b0, e0 = caps[0], caps[1]
                start = e0 if e0 > b0 else e0 + 1
                if start >= len(val):
                    break
            if out:
                return out
        return []


def test_device_algorithm_on_the_compiled_tables_reproduces_the_golden_vectors(golden):
    checked = skipped = 0
    for c in golden["regex"]:
        try:
            g = Grok(**c["config"])
        except GrokInitError:
            skipped += 1     # a Match entry neither device engine can run: Init fails loudly (no CPU path)
            continue
        t = TableGrok(g)
        for val, want in c["subs"]:
            checked += 1
            got = [[k, v.decode("latin-1")] for k, v in t.process_value(val.encode("latin-1"))]
            assert got == want, (c["config"]["Match"], val)
    assert checked >= 300 and skipped == 0


def test_prefix_screens_and_required_literals_are_necessary_conditions(golden, golden_dir):
    """Three prefilters sit in front of the NFA engine (lcGrokMatchDevice): the required literal, the TDFA screen of the
    pattern's prefix and the TDFA screen of the whole pattern, relaxed.  All must accept every value the pattern matches
    somewhere -- checked against the oracle's search on the golden values and on the configs[2] corpus; screens must exist
    for most NFA-engine patterns, and the relaxed one must reject values the prefix screen lets through."""
    from loongcollector_amd.grok_corpus import grok_lines
    with open(os.path.join(golden_dir, "grok_config3.json"), encoding="utf-8") as f:
        cfg3 = json.load(f)
    lib = Grok(CustomPatterns=cfg3["custom_patterns"])
    jobs = [(e, [v.encode("latin-1") for v, _ in c["subs"]]) for c in golden["regex"]
            for e in GrokOracle(c["config"]["Match"], custom_patterns=c["config"].get("CustomPatterns")).expanded]
    corpus = grok_lines(150)
    jobs += [(lib.denormalize(m), corpus) for m in cfg3["match"][4:40:3]]
    screens = checked = rejected = relaxed_screens = relaxed_only = 0
    for expanded, values in jobs:
        pat = expanded.encode("utf-8")
        try:
            rx = B.GpuRegex(pat, syntax_flags=GROK_SYNTAX)
        except B.RegexUnsupportedError:
            continue
        from oracle.oracle import ORX_NO_MOD_M, ORX_NO_MOD_S, ORX_REGEXP2, OracleRegex
        o = OracleRegex(pat, ORX_NO_MOD_S | ORX_NO_MOD_M | ORX_REGEXP2)
        lit = rx.required_literal()
        screen = B.GpuRegex.compile_screen(pat, syntax_flags=GROK_SYNTAX & ~B.LC_SYNTAX_SEARCH)
        it = TdfaInterp(screen) if screen is not None else None
        screens += screen is not None
        relaxed = B.GpuRegex.compile_screen(pat, syntax_flags=GROK_SYNTAX & ~B.LC_SYNTAX_SEARCH, max_states=20000,
                                            max_table_bytes=2 << 20, relaxed=True)
        it2 = TdfaInterp(relaxed) if relaxed is not None else None
        relaxed_screens += relaxed is not None
        for v in values:
            hit = o.search(v) is not None
            checked += 1
            passes = True
            if it is not None:
                passes = it.fullmatch(v) is not None
                assert passes or not hit, (expanded[:60], v)
                rejected += not passes
            if it2 is not None:
                passes2 = it2.fullmatch(v) is not None
                assert passes2 or not hit, ("relaxed", expanded[:60], v)
                relaxed_only += passes and not passes2
            if hit:
                assert lit in v, (expanded[:60], lit, v)
    assert checked >= 2000 and screens >= 15 and rejected > 300, (checked, screens, rejected)
    assert relaxed_screens >= 15 and relaxed_only > 50, (relaxed_screens, relaxed_only)


def test_tail_restamps_are_dead_stores_and_tail_states_collapse(golden_dir):
    """tdfa.cpp eliminateDeadStores + minimizeTdfaStates on a Grok format with a GREEDYDATA tail (CATALINALOG, a search pattern): as
    built, 95 % of the bytes of a matching line run a register program (the thread that would take over if the field ended here is
    re-derived, and re-stamped, at every byte) and the automaton has 1 410 states; none of those stamps is ever read.  After the two
    passes under 5 % of the bytes carry a program and the automaton fits the LDS window again -- with the same captures
    (test_regex_module_golden_vectors_on_the_tables and the GPU suite compare those)."""
    from loongcollector_amd.grok_corpus import grok_lines
    with open(os.path.join(golden_dir, "grok_config3.json"), encoding="utf-8") as f:
        cfg3 = json.load(f)
    lib = Grok(CustomPatterns=cfg3["custom_patterns"])
    pat = lib.denormalize("%{CATALINALOG}").encode("utf-8")
    rx = B.GpuRegex(pat, syntax_flags=GROK_SYNTAX, engine=B.LC_ENGINE_TDFA)
    assert rx.info()["states"] <= 200 and rx.table(B.LC_TABLE_TDFA_BLOB, np.uint32) is not None     # (LDS kernels, not L2)
    it = TdfaInterp(rx)
    lit = rx.required_literal()
    steps = programs = matched = 0
    for v in grok_lines(600):
        if lit not in v:
            continue
        state = it.start
        for b in v:
            t = int(it.trans[state * it.ncls + int(it.cmap[b])])
            steps += 1
            programs += (t >> 16) != 0
            state = t & 0xFFFF
            if state == 0:
                break
        matched += state != 0 and int(it.final_id[state]) != 0xFFFF
    assert matched >= 10 and steps > 10000 and programs < 0.05 * steps, (matched, steps, programs)


def test_literal_index_of_the_match_list(golden_dir):
    """The required literals of the 50-entry list as one Aho-Corasick DFA (grok_literal_index.cpp; the device walks it once per
    value: grok_literal_index_kernel): walked here exactly as the kernel does, bit p of the mask = "the value contains
    Match[p]'s literal" (always set for an entry without one) -- for every entry and every corpus line, plus adversarial values
    made of literal fragments (overlaps, a literal that is a suffix of another, repeated prefixes)."""
    from loongcollector_amd.grok_corpus import grok_lines
    with open(os.path.join(golden_dir, "grok_config3.json"), encoding="utf-8") as f:
        cfg3 = json.load(f)
    g = Grok(Match=cfg3["match"], CustomPatterns=cfg3["custom_patterns"], AnchoredFirst=False)
    blob = g.literal_index()
    assert blob is not None
    raw = blob.view(np.uint8)
    nstates, ncls, off_masks, off_table, always_lo, always_hi = [int(x) for x in blob[:6]]
    cmap = raw[32:288]
    masks = raw[off_masks:off_masks + 8 * nstates].view(np.uint64)
    table = raw[off_table:off_table + 2 * nstates * ncls].view(np.uint16)
    always = always_lo | (always_hi << 32)
    lits = []
    for i in range(g.n_match):
        rx = B.GpuRegex(g.expanded(i).encode("utf-8"), syntax_flags=GROK_SYNTAX)
        lits.append(rx.required_literal()[-32:])
    assert sum(1 for l in lits if l) >= 40 and always == sum(1 << i for i, l in enumerate(lits) if not l)

    def walk(v):
        state, mask = 0, always
        for b in v:
            e = int(table[state * ncls + int(cmap[b])])
            state = e & 0x7FFF
            if e & 0x8000:
                mask |= int(masks[state])
        return mask
    rng = random.Random(5)
    frags = [l[a:b] for l in lits if l for a in range(0, len(l), 3) for b in (a + 1, a + 4, len(l))]
    values = grok_lines(300) + [b"".join(rng.choice(frags) for _ in range(rng.randint(1, 12))) for _ in range(300)] + [b"", b" "]
    hits = 0
    for v in values:
        m = walk(v)
        for i, l in enumerate(lits):
            want = (not l) or (l in v)
            assert ((m >> i) & 1) == want, (i, l, v[:80])
            hits += bool(l) and want
    assert hits > 1000


@pytest.mark.parametrize("name, classes, slots", [("HTTPD_ERRORLOG", 68, 32), ("HAPROXYHTTP", 73, 106),
                                                  ("SYSLOGPAMSESSION", None, None), ("NAGIOSLOGLINE", 39, 278)])
def test_wide_table_formats_on_the_compiled_tables(golden_dir, name, classes, slots):
    """65..128 byte classes -> 4-word class masks; 65..128 / 129..320 capture slots -> 4 / 10 tag words per aux entry
    (device_tables.h NF_MASK_WORDS / NF_AUX_WORDS).  The device algorithm replayed on those tables must give the oracle's fields."""
    from tests.helpers.wide_patterns import wide_values
    with open(os.path.join(golden_dir, "grok_config3.json"), encoding="utf-8") as f:
        cfg3 = json.load(f)
    match = ["%{" + name + "}"]
    g = Grok(Match=match, CustomPatterns=cfg3["custom_patterns"])
    o = GrokOracle(match, custom_patterns=cfg3["custom_patterns"])
    t = TableGrok(g)
    it = t.interps[0]
    if classes:
        assert (it.ncls, it.nslots) == (classes, slots)
    else:
        assert len(it.runs) == 1      # "(?=%{GREEDYDATA:message})": the tables stamp the begin only
    matched = 0
    for v in wide_values():
        _, fields = o.process_value(v)
        assert t.process_value(v) == [[k, x] for k, x in fields], (name, v)
        matched += bool(fields)
    assert matched >= 4


def test_no_cpu_path():
    if B.load().lc_device_count() > 0:
        pytest.skip("a HIP device is present")
    g = Grok(Match=["%{WORD:w}"])
    with pytest.raises(B.GpuUnavailableError):
        g.match_host([b"abc"])
    with pytest.raises(B.GpuUnavailableError):
        g.process_logs([[("content", "abc")]])


def test_c_driven_oracle_walk_names_the_same_winners(golden_dir):
    """oracle/grok_baseline.c (the cpu_baseline leg of tools/grok_bench.py: processGrok driven from C) against the Python-driven walk
    of oracle/grok_oracle.py on generated values: the same values are won, by the same entry."""
    import json
    import numpy as np
    from loongcollector_amd.grok_corpus import grok_lines
    from oracle.grok_oracle import GrokOracle
    with open(os.path.join(golden_dir, "grok_config3.json"), encoding="utf-8") as f:
        cfg = json.load(f)
    o = GrokOracle(cfg["match"][:12], custom_patterns=cfg["custom_patterns"])
    vals = grok_lines(60)
    length = np.array([len(v) for v in vals], dtype=np.uint32)
    off = np.zeros(len(vals), dtype=np.uint32)
    off[1:] = np.cumsum(length[:-1])
    winner = o.first_match_batch(np.frombuffer(b"".join(vals), dtype=np.uint8), off, length)
    for v, w in zip(vals, winner):
        res, fields = o.process_value(v)
        assert (w >= 0) == (res == 0)
        if w >= 0:      # the winning entry is the first one whose own walk yields fields
            solo = GrokOracle([cfg["match"][int(w)]], custom_patterns=cfg["custom_patterns"])
            assert solo.process_value(v)[1] == fields


def test_configs2_corpus_has_the_baseline_shape(golden_dir):
    """BASELINE.json configs[2]: "mixed 128-4096B lines", 50 patterns.  Every line within 128..4096 bytes, lengths log-uniform (mean
    ~1.1 KB), and -- by the ORACLE's first-match-wins walk, not by assumption -- at least 35 of the 50 Match entries win some line
    (44 can: SYSLOGLINE / COMMONAPACHELOG shadow six entries in the reference's file order, loongcollector_amd/grok_corpus.py)."""
    import numpy as np
    from loongcollector_amd.grok_corpus import MAX_LINE, MIN_LINE, grok_lines
    with open(os.path.join(golden_dir, "grok_config3.json"), encoding="utf-8") as f:
        cfg3 = json.load(f)
    big = grok_lines(20000)
    lens = np.array([len(v) for v in big])
    assert lens.min() >= MIN_LINE == 128 and lens.max() <= MAX_LINE == 4096
    assert 1050 < lens.mean() < 1250 and 600 < np.median(lens) < 850          # log-uniform over 128..4096: mean 1145, median 724
    vals = big[:1500]
    o = GrokOracle(cfg3["match"], custom_patterns=cfg3["custom_patterns"])
    data = np.frombuffer(b"".join(vals), dtype=np.uint8)
    length = np.array([len(v) for v in vals], dtype=np.uint32)
    off = np.zeros(len(vals), dtype=np.uint32)
    off[1:] = np.cumsum(length[:-1])
    win = o.first_match_batch(data, off, length)
    winners = set(int(w) for w in win if w >= 0)
    assert len(winners) >= 35, sorted(winners)
    assert 0.02 < float((win < 0).mean()) < 0.09                               # ~5 % free text that no format takes


ANCHORED_IN_GLOBAL_MEMORY = ["%{HTTPD_ERRORLOG}", "%{CISCOFW106015}", "%{CISCOFW110002}", "%{CISCOFW402119}", "%{CISCOFW419001}",
                             "%{CISCOFW419002}", "%{CISCOFW710001_710002_710003_710005_710006}", "%{COMMONAPACHELOG}", "%{CISCOFW713172}"]


def test_anchored_automata_of_formats_that_searched_on_the_thread_list_engine(golden_dir):
    """Eight more Match entries of configs[2] get a tagged DFA for their anchored search since the construction stopped spending its
    commit budget on patterns without memberships and got a larger path budget (tdfa.cpp, regex_handle.cpp): HTTPD_ERRORLOG (the
    first entry of the list), six CISCOFW formats... -- 2 365..23 108 states (CISCOFW713172 needs 61 034 before minimisation: the
    table format's 16-bit state ids are the bound now), tables in global memory, walked by tdfa_wave_kernel.
    Until now these entries ran on the thread-list engine, so that engine's tables (same handle) are the reference here: on corpus
    lines the entry takes and on lines it does not, byte walk = wave walk = thread-list walk, captures included.  (End to end
    against the Grok oracle: tests/test_gpu_grok.py, on the device.)"""
    from loongcollector_amd.grok_corpus import grok_lines
    from tests.helpers.table_interp import NfaInterp, TdfaL2BlobInterp
    with open(os.path.join(golden_dir, "grok_config3.json"), encoding="utf-8") as f:
        cfg3 = json.load(f)
    g = Grok(Match=ANCHORED_IN_GLOBAL_MEMORY, CustomPatterns=cfg3["custom_patterns"], AnchoredFirst=False)
    flags = (B.LC_SYNTAX_SEARCH | B.LC_SYNTAX_NAMED_ONLY | B.LC_SYNTAX_NO_DOTALL | B.LC_SYNTAX_NO_MULTILINE | B.LC_SYNTAX_REGEXP2 |
             B.LC_SYNTAX_PREFIX)
    lines = grok_lines(3000, seed=77)
    taken = 0
    for i, name in enumerate(ANCHORED_IN_GLOBAL_MEMORY):
        rx = B.GpuRegex(g.expanded(i).encode(), syntax_flags=flags, engine=B.LC_ENGINE_TDFA)
        assert rx.table(B.LC_TABLE_TDFA_L2_BLOB, np.uint32) is not None and rx.has_nfa_program(), name
        l2 = TdfaL2BlobInterp(rx)
        assert 2000 < l2.nstates < 30000, (name, l2.nstates)
        res = [l2.fullmatch(l) for l in lines]
        hits = [k for k, r in enumerate(res) if r is not None]
        assert len(hits) >= 12, (name, len(hits))
        taken += len(hits)
        nfa = (AtomicNfaInterp if rx.atomic_groups()[0] else NfaInterp)(rx)
        for k in hits[:150] + [k for k, r in enumerate(res) if r is None][:150]:
            assert l2.fullmatch_wave(lines[k]) == res[k], (name, k)
            got = nfa.fullmatch(lines[k]) if rx.atomic_groups()[0] else nfa.fullmatch(lines[k], max_threads=128)
            assert got == res[k], (name, k, lines[k][:80])
    assert taken > 400
