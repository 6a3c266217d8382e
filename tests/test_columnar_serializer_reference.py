"""The serializer side of the bulk stitch (SURVEY.md section 8 f, rank 4) with the REFERENCE's own wire writer.

oracle/_ref/libref_processor.so holds core/protobuf/sls/LogGroupSerializer.cpp compiled from source: the hand-rolled protobuf writer the
SLS flusher uses, and its size functions.  Path A is what the agent does today: the reference's processor_parse_regex_native stitches K
(key, view) pairs into every LogEvent, then SLSEventGroupSerializer walks the events twice (CalculateLogEventSize, SerializeLogEvent:
SLSSerializer.cpp:254-268,377-395 -- the two loops are restated in the harness, the writer underneath is the reference's).  Path B is the
product's columnar hand-off: lc_processor_parse_columnar (csrc/c_processor_slot.cpp, here on tests/native/pipeline_double.cpp -- no GPU in
this container) returns one (begin, end) pair per event and key and content_bytes[i], the figure the serializer's first pass would have
computed; the same writer is fed from that table, no LogEvent holding the fields.

Checked: the two byte strings are the same -- every length prefix (content_bytes feeds StartToAddLog), every key and value, the order, the
nanosecond part.  CPU only; skipped where the reference tree is not present (the GPU box)."""
import ctypes
import json
import os
import random

import pytest

from test_reference_neighbours import RefPlugin
import test_pipeline_host_double as PD

REF = "/root/reference/core"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (/root/reference): its serializer is compiled from there")


class LcColumnar(ctypes.Structure):
    _fields_ = [("n_events", ctypes.c_uint32), ("n_keys", ctypes.c_uint32), ("keys", ctypes.POINTER(ctypes.c_char_p)),
                ("key_len", ctypes.POINTER(ctypes.c_uint32)), ("base", ctypes.POINTER(ctypes.c_void_p)), ("base_len", ctypes.POINTER(ctypes.c_uint32)),
                ("spans", ctypes.POINTER(ctypes.c_int32)), ("state", ctypes.POINTER(ctypes.c_uint8)), ("content_bytes", ctypes.POINTER(ctypes.c_uint64))]


@pytest.fixture(scope="module")
def libs():
    L = PD._double()            # the product: c_processor_slot.cpp + processor_parse_regex_gpu.cpp + event_model.cpp on the double
    vp, cp, sz = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t
    L.lc_processor_create.argtypes = [cp, ctypes.POINTER(vp), cp, sz]
    L.lc_processor_destroy.argtypes = [vp]
    L.lc_group_from_json.restype = vp
    L.lc_group_from_json.argtypes = [cp, cp, sz]
    L.lc_group_free.argtypes = [vp]
    L.lc_processor_parse_columnar.argtypes = [vp, vp, ctypes.POINTER(ctypes.POINTER(LcColumnar))]
    L.lc_columnar_free.argtypes = [ctypes.POINTER(LcColumnar)]
    R = RefPlugin.lib()
    R.refp_sls_serialize_group_json.restype = vp
    R.refp_sls_serialize_group_json.argtypes = [vp, cp, ctypes.c_int, ctypes.POINTER(sz), cp, sz]
    R.refp_sls_serialize_columnar.restype = vp
    R.refp_sls_serialize_columnar.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_uint32),
                                              ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint8),
                                              ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_int64),
                                              ctypes.c_int, ctypes.POINTER(sz)]
    return L, R


LINES = ["GET /index.html 200 curl/8.1", "POST /api/v1/x 404 Mozilla/5.0 (X11; Linux x86_64)", "nomatch", "", "HEAD / 204 ", "GET /café 200 ünï",
         "GET /" + "a" * 300 + " 200 " + "u" * 200, "DELETE /x 500 -"]     # (values beyond 127 bytes: two-byte length prefixes)
CONFIGS = [
    {"SourceKey": "content", "Regex": r"(\w+) (\S+) (\d{3}) (.*)", "Keys": ["method", "path", "status", "ua"]},
    {"SourceKey": "content", "Regex": r"(\w+) (\S+) (\d{3})(?: (\S+))?.*", "Keys": ["method", "path", "status", "first_ua_word"]},   # a group that may not take part
    {"SourceKey": "content", "Regex": r"(\w+) (.*)", "Keys": ["m", "a_key_name_that_is_longer_than_the_usual_ones_" + "k" * 100]},
]


@pytest.mark.parametrize("enable_ns", [0, 1])
def test_the_reference_s_wire_writer_fed_from_the_columnar_table(libs, enable_ns):
    L, R = libs
    rng = random.Random(5 + enable_ns)
    sz = ctypes.c_size_t
    total = 0
    for config in CONFIGS:
        ref = RefPlugin("processor_parse_regex_native", config)
        h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(512)
        assert L.lc_processor_create(json.dumps(config).encode(), ctypes.byref(h), err, 512) == 0, err.value
        for _ in range(40):
            n = rng.randint(0, 30)
            events = []
            for k in range(n):
                ev = {"contents": {"content": rng.choice(LINES)}, "timestamp": 1700000000 + k, "type": 1}
                if rng.random() < 0.5:
                    ev["timestampNanosecond"] = rng.randrange(10 ** 9)
                events.append(ev)
            text = json.dumps({"events": events}).encode()
            # path A: the reference's processor, then its serializer's two passes over the stitched events
            alen = sz()
            a = R.refp_sls_serialize_group_json(ref.h, text, enable_ns, ctypes.byref(alen), err, 512)
            assert a, err.value
            a_bytes = ctypes.string_at(a, alen.value)
            R.refp_free(a)
            # path B: the product's columnar table into the same writer
            g = L.lc_group_from_json(text, err, 512)
            assert g, err.value
            col = ctypes.POINTER(LcColumnar)()
            assert L.lc_processor_parse_columnar(h, g, ctypes.byref(col)) == 0
            c = col.contents
            assert c.n_events == n
            ts = (ctypes.c_uint32 * max(1, n))(*[e["timestamp"] for e in events])
            ns = (ctypes.c_int64 * max(1, n))(*[e.get("timestampNanosecond", -1) for e in events])
            blen = sz()
            b = R.refp_sls_serialize_columnar(c.n_events, c.n_keys, c.keys, c.key_len, c.base, c.spans, c.state, c.content_bytes, ts, ns, enable_ns,
                                              ctypes.byref(blen))
            b_bytes = ctypes.string_at(b, blen.value)
            R.refp_free(b)
            parsed = sum(1 for i in range(n) if c.state[i] == 1)
            L.lc_columnar_free(col)
            L.lc_group_free(g)
            assert a_bytes == b_bytes, (config, events)
            assert (len(a_bytes) > 0) == (parsed > 0)
            total += len(a_bytes)
        L.lc_processor_destroy(h)
    assert total > 100000
