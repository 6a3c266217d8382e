"""The HOST translation unit of processor_filter_regex_gpu (csrc/processor_filter_gpu.cpp: Init's precedence of Include / FilterKey +
FilterRegex / ConditionExp and what it refuses, the expression tree, which events survive, the non-UTF-8 blanking) on a machine without a
GPU, beside the REFERENCE's own ProcessorFilterNative.cpp compiled from /root/reference (oracle/_ref/libref_processor.so).

The product's code talks to the HIP runtime itself; tests/native/filter_double.cpp stands in for it on the CPU ("device" memory is host
memory, the one match call per group is answered by the oracle's regex).  Compared: for randomly GENERATED configs -- all three rule
forms, several at once, malformed ones -- whether Init accepts, and on random event groups the events left, every field, in order.
CPU only; skipped where the reference tree is not present (the GPU box)."""
import ctypes
import json
import os
import random
import subprocess

import pytest

from test_reference_neighbours import RefPlugin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/core"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (/root/reference): its filter is compiled from there")

_LIB = None


def _double():
    global _LIB
    if _LIB is not None:
        return _LIB
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libfilter_double.so")
    csrc = os.path.join(ROOT, "loongcollector_amd", "csrc")
    srcs = [os.path.join(ROOT, "tests", "native", "filter_double.cpp"), os.path.join(csrc, "processor_filter_gpu.cpp"), os.path.join(csrc, "event_model.cpp")]
    deps = srcs + [os.path.join(csrc, h) for h in ("event_model.hpp", "processor_filter_gpu.hpp", "json_min.hpp", "trip_buffers.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-w", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include",
                               "-I", os.path.join(ROOT, "include"), "-I", csrc, "-o", so] + srcs +
                              ["-L" + os.path.join(ROOT, "oracle"), "-loracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,--no-undefined", "-Wl,-Bsymbolic"])
    L = ctypes.CDLL(so)
    vp, cp, sz = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t
    L.lc_filter_create.argtypes = [cp, ctypes.POINTER(vp), cp, sz]
    L.lc_filter_destroy.argtypes = [vp]
    L.lc_filter_mode.argtypes = [vp]
    L.lc_filter_counters.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.fd_process_json.restype = vp
    L.fd_process_json.argtypes = [vp, cp, cp, sz]
    L.fd_free.argtypes = [vp]
    _LIB = L
    return L


def _whole(text):
    d = json.loads(text, object_pairs_hook=list)
    out = []
    for ev in dict(d or []).get("events", []):
        ev = dict(ev)
        ev["contents"] = [tuple(kv) for kv in ev.get("contents", [])]
        out.append(sorted(ev.items(), key=lambda kv: kv[0]))
    return out


class ProductFilter:
    def __init__(self, config):
        self.L = _double()
        self.h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(512)
        if self.L.lc_filter_create(json.dumps(config).encode(), ctypes.byref(self.h), err, 512) != 0:
            self.h = None
            raise ValueError(err.value.decode("utf-8", "replace"))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.lc_filter_destroy(self.h)
            self.h = None

    def mode(self):
        return self.L.lc_filter_mode(self.h)

    def process(self, fixture_bytes):
        err = ctypes.create_string_buffer(512)
        p = self.L.fd_process_json(self.h, fixture_bytes, err, 512)
        assert p, err.value
        try:
            return _whole(ctypes.string_at(p).decode("latin-1"))
        finally:
            self.L.fd_free(p)

    def counters(self):
        c = (ctypes.c_uint64 * 2)()
        self.L.lc_filter_counters(self.h, c)
        return (int(c[0]), int(c[1]))


def _ref_process(ref, fixture_bytes):
    err = ctypes.create_string_buffer(512)
    p = ref.L.refp_process_json(ref.h, fixture_bytes, err, 512)
    assert p, err.value
    try:
        return _whole(ctypes.string_at(p).decode("latin-1"))
    finally:
        ref.L.refp_free(p)


KEYS = ["status", "method", "path", "ua", "ip", "absent"]
REGEXES = ["2\\d\\d|30[14]", "GET|HEAD", "/api/.*", ".*(?:bot|curl).*", "10\\.\\d+\\.\\d+\\.\\d+", "[23]\\d\\d", "/admin(?:/.*)?", ".*", "", "(?i)get", "\\d+",
           ".*caf.*", "[a-z/]+", "("]          # (the last one is no regex)
FIELDS = {
    "status": ["200", "204", "301", "404", "500", "2000", ""],
    "method": ["GET", "HEAD", "POST", "GETX", "get"],
    "path": ["/api/v1/x", "/admin", "/admin/users", "/index.html", "/apix", "/café", "/bad\\xff\\xfebytes", "/\\xe2\\x82truncated"],
    "ua": ["curl/8.1", "Mozilla/5.0", "Googlebot/2.1", "bot", ""],
    "ip": ["10.0.0.1", "192.168.1.1", "10.1.2.3.4", "10.x.0.1"],
}


def _leaf(rng):
    kind = rng.random()
    # (a ConditionExp leaf whose `exp` is no regex is not generated: the reference constructs boost::regex in RegexFilterValueNode's
    # initialiser list, ProcessorFilterNative.h:70-71, and the exception leaves Init -- the product refuses the config with a message)
    node = {"type": "regex", "key": rng.choice(KEYS), "exp": rng.choice(REGEXES[:-1])}
    if kind < 0.04:
        node["type"] = rng.choice(["equals", "", 3])
    elif kind < 0.07:
        del node[rng.choice(["type", "key", "exp"])]
    elif kind < 0.09:
        node[rng.choice(["key", "exp"])] = rng.choice([5, None, ["x"]])
    return node


def _tree(rng, depth):
    if depth == 0 or rng.random() < 0.35:
        return _leaf(rng)
    op = rng.choice(["and", "or", "not", "AND", "Or", "nOt"] if rng.random() < 0.9 else ["xor", "", 1])
    want = 1 if str(op).lower() == "not" else 2
    n = want if rng.random() < 0.9 else rng.choice([0, 1, 2, 3])
    node = {"operator": op, "operands": [_tree(rng, depth - 1) for _ in range(n)]}
    if rng.random() < 0.03:
        node["operands"] = "no array"
    if rng.random() < 0.03:
        del node["operator"]
    return node


def _config(rng):
    config = {}
    forms = rng.choice([["Include"], ["FilterKey"], ["ConditionExp"], ["ConditionExp"], ["Include", "ConditionExp"], ["FilterKey", "Include"],
                        ["FilterKey", "ConditionExp"], ["Include", "FilterKey", "ConditionExp"], []])
    if "Include" in forms:
        inc = {rng.choice(KEYS): rng.choice(REGEXES) for _ in range(rng.randint(0, 3))}
        config["Include"] = inc if rng.random() < 0.92 else rng.choice(["str", ["a"], {"k": 3}, None])
    if "FilterKey" in forms:
        n = rng.randint(0, 3)
        config["FilterKey"] = [rng.choice(KEYS) for _ in range(n)]
        config["FilterRegex"] = [rng.choice(REGEXES) for _ in range(n if rng.random() < 0.85 else rng.randint(0, 3))]
        if rng.random() < 0.08:
            config[rng.choice(["FilterKey", "FilterRegex"])] = rng.choice(["str", [1, 2], None])
        if rng.random() < 0.05:
            del config["FilterRegex"]
    if "ConditionExp" in forms:
        config["ConditionExp"] = _tree(rng, 3) if rng.random() < 0.95 else rng.choice(["str", [], None, 7])
    if rng.random() < 0.4:
        config["DiscardingNonUTF8"] = rng.choice([True, True, False, "yes"])
    return config


def _group_bytes(rng, n):
    events = []
    for k in range(n):
        e = [[key, rng.choice(vals)] for key, vals in FIELDS.items() if rng.random() < 0.8]
        rng.shuffle(e)
        ev = {"contents": e, "timestamp": 1700000000 + k, "type": 1}
        if rng.random() < 0.3:
            ev["timestampNanosecond"] = k
        events.append(ev)
    # (bytes that are no UTF-8 travel as literal "\\xNN" through the JSON writer and become the bytes themselves behind it)
    text = json.dumps({"events": events}, ensure_ascii=False).encode("utf-8")
    for esc, raw in ((b"\\\\xff", b"\xff"), (b"\\\\xfe", b"\xfe"), (b"\\\\xe2", b"\xe2"), (b"\\\\x82", b"\x82")):
        text = text.replace(esc, raw)
    return text


def test_generated_configs_init_and_filter_like_the_reference():
    rng = random.Random(20260922)
    accepted = refused = events_seen = kept = 0
    modes = set()
    for trial in range(1500):
        config = _config(rng)
        try:
            ref = RefPlugin("processor_filter_regex_native", config)
        except ValueError:
            ref = None
        try:
            prod = ProductFilter(config)
        except ValueError:
            prod = None
        assert (ref is None) == (prod is None), (config, "reference refuses" if ref is None else "reference accepts")
        if ref is None:
            refused += 1
            continue
        accepted += 1
        modes.add(prod.mode())
        for _ in range(2):
            g = _group_bytes(rng, rng.randint(1, 25))
            got, want = prod.process(g), _ref_process(ref, g)
            assert got == want, (config, g)
            events_seen += len(json.loads(g.decode("latin-1"))["events"])
            kept += len(got)
    assert accepted > 600 and refused > 200 and modes == {0, 1, 2}
    assert 0.05 * events_seen < kept < 0.95 * events_seen


def test_the_imported_unit_test_vectors_through_the_product_s_host_code(golden_dir):
    with open(os.path.join(golden_dir, "filter_vectors.json"), encoding="utf-8") as f:
        vectors = json.load(f)
    for c in vectors["cases"]:
        if not c["in"] or any(not e for e in c["in"]):
            continue
        p = ProductFilter(c["config"])
        g = json.dumps({"events": [{"contents": e, "timestamp": 1, "type": 1} for e in c["in"]]}).encode()
        assert [dict(dict(ev)["contents"]) for ev in p.process(g)] == c["out"], c["cite"]
        assert p.counters() == (len(c["in"]), len(c["out"]))
    for c in vectors["init_fail"]:
        with pytest.raises(ValueError):
            ProductFilter(c["config"])
    for c in vectors["init_ok"]:
        ProductFilter(c["config"])


def test_the_trip_s_completion_word_is_never_spelled_by_an_earlier_group_s_status_bytes():
    """the zero-copy trip's completion word lay BEHIND the status bytes, at roundup64(values), so it moved with
    the group's size; a shorter group found there the status bytes of an earlier, longer one -- and bytes (1,1,0,0) read as a word are
    257, trip 257's own number: the host's wait returned before the device had run.  The double's device is asynchronous (queued work
    runs when the host waits, and a wait that finds its number already there returns at once and drops the work), its pinned memory is
    handed out dirty; groups alternate between 200 and 64 events over 600 trips, the long ones carrying at [64..68) every byte pattern
    that spells one of the coming trip numbers."""
    L = _double()
    L.fd_early_returns.restype = ctypes.c_uint64
    before = L.fd_early_returns()
    prod = ProductFilter({"Include": {"k": "yes.*"}})
    rng = random.Random(7)

    def group(flags):
        return json.dumps({"events": [{"contents": [["k", ("yes%d" if f else "no%d") % i]], "timestamp": i, "type": 1}
                                      for i, f in enumerate(flags)]}).encode()

    trip = 0
    for _ in range(300):
        nxt = trip + 2                                   # the SHORT group's trip number
        flags = [rng.random() < 0.5 for _ in range(200)]
        for b in range(4):                               # statuses 64..67 of the long group = the bytes of `nxt`, where they are 0 / 1
            byte = (nxt >> (8 * b)) & 0xFF
            flags[64 + b] = byte == 1
        for flags_now in (flags, [rng.random() < 0.5 for _ in range(64)]):
            trip += 1
            got = prod.process(group(flags_now))
            want = [i for i, f in enumerate(flags_now) if f]
            assert [int(dict(ev)["contents"][0][1][3:]) for ev in got] == want, trip
    assert trip == 600 and L.fd_early_returns() == before
