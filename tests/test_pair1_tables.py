"""ONE-STAMP byte-pair tables (regex_handle.cpp planTdfaDerive + packTdfaBlob pairMode 2; device_tables.h TP1_*): what the compact
512-lane kernel walks by default for small automata since round 3.  The walk of tdfaStreamPair1Chunk / tdfaSettleDoubles is restated
store for store in tests/helpers/table_interp.py TdfaPair1Interp; here it is pinned against the single-byte walk of the same blob and
against the oracle, for every alignment of the line in memory (pairs and chunks are aligned in memory, not in the line).  The kernel
itself is compared with the oracle in the GPU suite (and was, bit-exact, in profiles/round3_lab_pair1.txt)."""
import json
import os

import numpy as np
import pytest

from loongcollector_amd import binding as B, corpus
from oracle.oracle import OracleRegex
from tests.helpers.table_interp import TdfaBlobInterp, TdfaPair1Interp

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _pair1(rx):
    blob = rx.table(B.LC_TABLE_TDFA_WIDE_BLOB, np.uint32)
    if blob is None or not int(blob[7]):
        return None
    po = int(blob[7]) // 4
    return TdfaPair1Interp(rx) if int(blob[po + 4]) == 1 else None


def test_default_tables_of_the_headline_regex_are_one_stamp_pair_tables():
    rx = B.GpuRegex(corpus.REGEX_A)
    blob = rx.table(B.LC_TABLE_TDFA_WIDE_BLOB, np.uint32)
    assert int(blob[15]) == 512 and int(blob[7]) != 0           # TD_BLOCK, TD_OFF_PAIR
    it = _pair1(rx)
    assert it is not None
    # end of field k on the separator, start of field k+1 on the byte behind it: the start registers are derived, not stamped
    assert sorted((b, a, d) for b, a, d in it.derive) == [(6, 5, 1), (10, 9, 1), (12, 11, 1), (14, 13, 1)]
    # the benchmark regex (1.8 % DOUBLE entries: fields that may be empty on both sides of a separator) takes them too since round 4
    rxb = B.GpuRegex(corpus.REGEX_B)
    blob_b = rxb.table(B.LC_TABLE_TDFA_WIDE_BLOB, np.uint32)
    assert int(blob_b[15]) == 512 and int(blob_b[7]) != 0 and _pair1(rxb) is not None
    assert blob_b.nbytes + 33 * 512 * 2 + 8 * 64 * 64 <= 80 * 1024          # two workgroups per CU (tables + 16-bit registers + tiles)
    # the standard tables (small batches, the in-agent shape, the multi-tenant launch) carry them too since round 4, at the workgroup
    # size they had without (two workgroups per CU: tables + 32-bit registers + padded tiles <= 80 KiB); search patterns do not
    for r in (rx, rxb):
        std = r.table(B.LC_TABLE_TDFA_BLOB, np.uint32)
        po = int(std[7]) // 4
        assert po and int(std[po + 4]) == 1 and int(std[15]) == 256
    srch = B.GpuRegex(corpus.REGEX_A, syntax_flags=B.LC_SYNTAX_SEARCH).table(B.LC_TABLE_TDFA_BLOB, np.uint32)
    assert int(srch[7]) == 0


@pytest.mark.parametrize("kind", ["A", "B"])
def test_pair_walk_equals_single_byte_walk_and_oracle_on_the_bench_corpus(kind, monkeypatch):
    if kind == "B":
        monkeypatch.setenv("LC_TDFA_PAIR", "2")                 # (B does not take the pair table by default: forced, it must still be exact)
        monkeypatch.setenv("LC_TDFA_COMPACT", "512")
    pattern = corpus.REGEX_A if kind == "A" else corpus.REGEX_B
    rx = B.GpuRegex(pattern)
    p1, plain = _pair1(rx), TdfaBlobInterp(rx, compact=True)
    assert p1 is not None
    data, off, length = corpus.apache_batch(160, kind, poison_every=7, empty_every=3)
    exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off[:-1], length)
    lines = [bytes(data[off[i]:off[i] + length[i]]) for i in range(160)]
    lines += [b"", b"x", b'1 - - [t z] "G /" 0 1 2 - "-" "-"', b'1.1 a b [x y] "GET " 0.1 12 200 5 "" ""',
              b'9 - - [a b] "P u" 1 2 3 4 "r" "b"', b'a b c [d] "e f g" h i "j" "k"', b'a b c [] " " h i "" ""']
    doubles = 0
    for i, s in enumerate(lines):
        want = plain.fullmatch(s)
        if i < 160:
            assert (want is not None) == bool(exp_status[i]) and (want is None or want == [int(v) for v in exp_caps[i]])
        for head in range(16):
            assert p1.fullmatch_pair1(s, head=head) == want, (kind, i, head, s[:60])
            doubles += p1.doubles
    assert doubles > 0                                          # (the settled-double path ran)


def test_pair_walk_on_the_golden_patterns_that_take_a_pair_table(monkeypatch):
    """every golden full-match pattern whose automaton takes a one-stamp pair table when asked to (small, no general register
    program): the pair walk against the vectors' captures (PCRE1 and CPython re agree on them), three alignments each"""
    monkeypatch.setenv("LC_TDFA_PAIR", "2")
    monkeypatch.setenv("LC_TDFA_COMPACT", "512")
    with open(os.path.join(GOLDEN, "regex_golden.json"), encoding="utf-8") as f:
        golden = json.load(f)
    took = checked = 0
    for c in golden["cases"]:
        try:
            rx = B.GpuRegex(c["p"].encode("latin-1"))
        except B.RegexUnsupportedError:
            continue
        if rx.info()["engine"] != B.LC_ENGINE_TDFA:
            continue
        p1 = _pair1(rx)
        if p1 is None:
            continue
        took += 1
        for sub, want in c["subs"][:10]:
            s = sub.encode("latin-1")
            exp = None if want is None else [int(v) for v in want[2:]]      # (the vectors carry group 0 first)
            for head in (0, 5, 15):
                assert p1.fullmatch_pair1(s, head=head) == exp, (c["p"], s[:40], head)
                checked += 1
    assert took >= 20 and checked >= 500, (took, checked)


def test_standard_tables_pack_the_same_pair_table_when_asked(monkeypatch):
    """LC_TDFA_PAIR=2 without LC_TDFA_COMPACT: the standard tables (32-bit registers: small batches, the in-agent shape) carry the
    one-stamp pair table too -- prepared for the next round, walked here on the CPU"""
    monkeypatch.setenv("LC_TDFA_PAIR", "2")
    rx = B.GpuRegex(corpus.REGEX_A)
    std = rx.table(B.LC_TABLE_TDFA_BLOB, np.uint32)
    assert int(std[7]) != 0 and int(std[int(std[7]) // 4 + 4]) == 1
    p1, plain = TdfaPair1Interp(rx, compact=False), TdfaBlobInterp(rx)
    data, off, length = corpus.apache_batch(40, "A", poison_every=5, empty_every=3)
    for i in range(40):
        s = bytes(data[off[i]:off[i] + length[i]])
        for head in (0, 3, 8, 13):
            assert p1.fullmatch_pair1(s, head=head) == plain.fullmatch(s)


def test_differential_fuzz_of_the_pair_tables_short():
    """tools/fuzz_pair1.py on two seeds (the long run -- 100 seeds, 12 698 tables, 355 538 checks -- is in
    profiles/round3_tdfa_experiments.txt): random patterns, full-match and search mode, random alignments, against the oracle"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_pair1.py"), "200", "202"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("ok:"), (out.stdout[-400:], out.stderr[-1200:])


@pytest.mark.parametrize("kind", ["A", "B"])
def test_pair_walk_on_the_standard_tables(kind):
    """Round 4: the STANDARD (32-bit register) tables carry the one-stamp pair table by default; the pair walk over them against the
    single-byte walk of the same blob and the oracle, at several alignments of the line in memory."""
    pattern = corpus.REGEX_A if kind == "A" else corpus.REGEX_B
    rx = B.GpuRegex(pattern)
    p1, plain = TdfaPair1Interp(rx, compact=False), TdfaBlobInterp(rx)
    data, off, length = corpus.apache_batch(120, kind, poison_every=7, empty_every=3)
    exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off[:-1], length)
    for i in range(120):
        s = bytes(data[off[i]:off[i] + length[i]])
        want = plain.fullmatch(s)
        assert (want is not None) == bool(exp_status[i]) and (want is None or want == [int(v) for v in exp_caps[i]])
        for head in (0, 3, 7, 15):
            assert p1.fullmatch_pair1(s, head=head) == want, (kind, i, head)
