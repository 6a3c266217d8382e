"""ctypes binding of the processor layer (include/lc_processor.h): create a processor_parse_regex_gpu instance from
its JSON plugin config, run it over event groups given in the reference unit tests' JSON fixture format."""
import ctypes
import json

from . import binding

COUNTER_NAMES = ["discarded_events_total", "out_failed_events_total", "out_key_not_found_events_total",
                 "out_successful_events_total", "in_events_total", "out_events_total", "in_size_bytes",
                 "out_size_bytes", "total_process_time_us", "complexity_exceeded_events_total", "undecided_events_total",
                 "device_failed_events_total"]


class LcColumnar(ctypes.Structure):
    _fields_ = [("n_events", ctypes.c_uint32), ("n_keys", ctypes.c_uint32), ("keys", ctypes.POINTER(ctypes.c_char_p)),
                ("key_len", ctypes.POINTER(ctypes.c_uint32)), ("base", ctypes.POINTER(ctypes.c_void_p)),
                ("base_len", ctypes.POINTER(ctypes.c_uint32)), ("spans", ctypes.POINTER(ctypes.c_int32)),
                ("state", ctypes.POINTER(ctypes.c_uint8)), ("content_bytes", ctypes.POINTER(ctypes.c_uint64))]


class ProcessorInitError(ValueError):
    pass


def _lib():
    L = binding.load()
    if not getattr(L, "_lc_processor_bound", False):
        vp, cp, sz = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t
        L.lc_processor_create.restype = ctypes.c_int
        L.lc_processor_create.argtypes = [cp, ctypes.POINTER(vp), cp, sz]
        L.lc_processor_destroy.argtypes = [vp]
        L.lc_processor_key_count.argtypes = [vp]
        L.lc_processor_key.restype = cp
        L.lc_processor_key.argtypes = [vp, ctypes.c_int]
        L.lc_processor_process.restype = ctypes.c_int
        L.lc_processor_process.argtypes = [vp, vp]
        L.lc_processor_counters.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
        L.lc_processor_set_alarm_sink.restype = None
        L.lc_processor_set_alarm_sink.argtypes = [vp, vp, vp]
        L.lc_filter_create.restype = ctypes.c_int
        L.lc_filter_create.argtypes = [cp, ctypes.POINTER(vp), cp, sz]
        L.lc_filter_destroy.argtypes = [vp]
        L.lc_filter_mode.argtypes = [vp]
        L.lc_filter_process.restype = ctypes.c_int
        L.lc_filter_process.argtypes = [vp, vp]
        L.lc_filter_counters.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
        L.lc_filter_none_utf8.restype = ctypes.c_int
        L.lc_filter_none_utf8.argtypes = [cp, sz, ctypes.c_int]
        L.lc_group_native.restype = vp
        L.lc_group_native.argtypes = [vp]
        L.lc_group_from_json.restype = vp
        L.lc_group_from_json.argtypes = [cp, cp, sz]
        L.lc_processor_parse_columnar.restype = ctypes.c_int
        L.lc_processor_parse_columnar.argtypes = [vp, vp, ctypes.POINTER(ctypes.POINTER(LcColumnar))]
        L.lc_columnar_free.argtypes = [ctypes.POINTER(LcColumnar)]
        L.lc_group_from_lines.restype = vp
        L.lc_group_from_lines.argtypes = [vp, vp, vp, ctypes.c_uint32, cp]
        L.lc_group_to_json.restype = vp
        L.lc_group_to_json.argtypes = [vp]
        L.lc_group_event_count.restype = sz
        L.lc_group_event_count.argtypes = [vp]
        L.lc_group_free.argtypes = [vp]
        L.lc_free.argtypes = [vp]
        L.lc_group_from_buffer.restype = vp
        L.lc_group_from_buffer.argtypes = [vp, sz, cp, ctypes.c_uint64, cp]
        L.lc_pipeline_create.restype = ctypes.c_int
        L.lc_pipeline_create.argtypes = [cp, ctypes.POINTER(vp), cp, sz]
        L.lc_pipeline_destroy.argtypes = [vp]
        L.lc_pipeline_is_fused.restype = ctypes.c_int
        L.lc_pipeline_is_fused.argtypes = [vp]
        L.lc_pipeline_process.restype = ctypes.c_int
        L.lc_pipeline_process.argtypes = [vp, vp]
        L.lc_pipeline_counters.restype = ctypes.c_int
        L.lc_pipeline_counters.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
        L._lc_processor_bound = True
    return L


class EventGroup:
    """A PipelineEventGroup built from the fixture JSON ({"events":[{"contents":{..},"timestamp":..,"type":1}]})."""

    def __init__(self, fixture):
        text = fixture if isinstance(fixture, str) else json.dumps(fixture)
        err = ctypes.create_string_buffer(256)
        self._L = _lib()
        self._h = self._L.lc_group_from_json(text.encode("utf-8"), err, 256)
        if not self._h:
            raise ValueError(err.value.decode())

    @classmethod
    def from_lines(cls, data, off, length, key="content"):
        """The group a file input hands over: n lines (numpy: data u8[], off u32[n], length u32[n]) copied once into the group's
        SourceBuffer, one log event per line with `key` = a view of its line (lc_group_from_lines)."""
        import numpy as np
        data = np.ascontiguousarray(data, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        length = np.ascontiguousarray(length, dtype=np.uint32)
        self = cls.__new__(cls)
        self._L = _lib()
        self._h = self._L.lc_group_from_lines(data.ctypes.data, off.ctypes.data, length.ctypes.data, len(off), key.encode("utf-8"))
        if not self._h:
            raise ValueError("lc_group_from_lines failed")
        return self

    @classmethod
    def from_buffer(cls, buf, key="content", file_offset=0, file_offset_key=None):
        """The group BEFORE the line splitter: one log event holding a copy of the read buffer (lc_group_from_buffer)."""
        self = cls.__new__(cls)
        self._L = _lib()
        raw = bytes(buf)
        self._h = self._L.lc_group_from_buffer(raw, len(raw), key.encode("utf-8"), file_offset,
                                               None if file_offset_key is None else file_offset_key.encode("utf-8"))
        if not self._h:
            raise ValueError("lc_group_from_buffer failed")
        return self

    def to_json(self):
        p = self._L.lc_group_to_json(self._h)
        try:
            return ctypes.string_at(p).decode("utf-8")
        finally:
            self._L.lc_free(p)

    def to_dict(self):
        return json.loads(self.to_json())

    def contents(self):
        """-> per event: ordered list of (key, value) (None for non-log events)"""
        out = []
        d = json.loads(self.to_json(), object_pairs_hook=list)
        events = dict(d).get("events", [])
        for ev in events:
            ev = dict(ev)
            out.append(list(ev.get("contents", [])) if ev.get("type") == 1 else None)
        return out

    def __len__(self):
        return int(self._L.lc_group_event_count(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.lc_group_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Processor:
    """processor_parse_regex_gpu; same config keys as processor_parse_regex_native."""

    def __init__(self, config):
        text = config if isinstance(config, str) else json.dumps(config)
        self._L = _lib()
        h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(512)
        rc = self._L.lc_processor_create(text.encode("utf-8"), ctypes.byref(h), err, 512)
        if rc != binding.LC_OK:
            raise ProcessorInitError(err.value.decode())
        self._h = h

    @property
    def keys(self):
        n = self._L.lc_processor_key_count(self._h)
        return [self._L.lc_processor_key(self._h, i).decode() for i in range(n)]

    def process(self, group: EventGroup):
        rc = self._L.lc_processor_process(self._h, group._h)
        if rc == binding.LC_ERR_NO_DEVICE:
            raise binding.GpuUnavailableError("processor_parse_regex_gpu: no usable HIP device (no CPU path)")
        if rc != binding.LC_OK:
            raise RuntimeError("lc_processor_process rc=%d" % rc)

    def parse_columnar(self, group: EventGroup):
        """lc_processor_parse_columnar: the capture table of the group next to the values' base pointers, nothing stitched.
        -> list per event: None (skipped), False (parse failure) or ([(key, value bytes)], content_bytes)"""
        c = ctypes.POINTER(LcColumnar)()
        rc = self._L.lc_processor_parse_columnar(self._h, group._h, ctypes.byref(c))
        if rc != binding.LC_OK:
            raise RuntimeError("lc_processor_parse_columnar rc=%d" % rc)
        try:
            col = c.contents
            K = col.n_keys
            keys = [col.keys[k].decode() for k in range(K)]
            out = []
            for i in range(col.n_events):
                if col.state[i] == 0:
                    out.append(None)
                elif col.state[i] == 2:
                    out.append(False)
                else:
                    raw = ctypes.string_at(col.base[i], col.base_len[i])
                    fields = []
                    for k in range(K):
                        b, e = col.spans[(i * K + k) * 2], col.spans[(i * K + k) * 2 + 1]
                        fields.append((keys[k], b"" if b < 0 else raw[b:e]))
                    out.append((fields, int(col.content_bytes[i])))
            return out
        finally:
            self._L.lc_columnar_free(c)

    def parse_columnar_count(self, group: EventGroup):
        """the same call without building Python objects (throughput measurements) -> (events, parsed events, content bytes)"""
        c = ctypes.POINTER(LcColumnar)()
        rc = self._L.lc_processor_parse_columnar(self._h, group._h, ctypes.byref(c))
        if rc != binding.LC_OK:
            raise RuntimeError("lc_processor_parse_columnar rc=%d" % rc)
        try:
            col = c.contents
            n = int(col.n_events)
            import numpy as np
            state = np.ctypeslib.as_array(col.state, shape=(n,)) if n else np.zeros(0, np.uint8)
            sizes = np.ctypeslib.as_array(col.content_bytes, shape=(n,)) if n else np.zeros(0, np.uint64)
            return n, int((state == 1).sum()), int(sizes.sum())
        finally:
            self._L.lc_columnar_free(c)

    def parse_columnar_discard(self, group: EventGroup):
        """lc_processor_parse_columnar + lc_columnar_free and nothing else: the call as a serializer thread would make it, without the
        ctypes -> numpy conversions of parse_columnar_count (60-80 us per group under the GIL: with 16 threads THEY were what the
        bench leg measured)"""
        c = ctypes.POINTER(LcColumnar)()
        rc = self._L.lc_processor_parse_columnar(self._h, group._h, ctypes.byref(c))
        if rc != binding.LC_OK:
            raise RuntimeError("lc_processor_parse_columnar rc=%d" % rc)
        self._L.lc_columnar_free(c)

    def collect_alarms(self):
        """lc_processor_set_alarm_sink: -> the list that receives (kind, message bytes) for every REGEX_MATCH_ALARM the
        reference would raise (ProcessorParseRegexNative.cpp:196-244)"""
        out = []
        proto = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char), ctypes.c_size_t)
        self._alarm_cb = proto(lambda user, kind, msg, n: out.append((kind, ctypes.string_at(msg, n))))
        self._L.lc_processor_set_alarm_sink(self._h, ctypes.cast(self._alarm_cb, ctypes.c_void_p), None)
        return out

    def counters(self):
        buf = (ctypes.c_uint64 * len(COUNTER_NAMES))()
        self._L.lc_processor_counters(self._h, buf)
        return dict(zip(COUNTER_NAMES, [int(x) for x in buf]))

    def close(self):
        if getattr(self, "_h", None):
            self._L.lc_processor_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


PIPE_COUNTER_NAMES = ["filter_in_events", "filter_out_events", "groups_fused", "groups_chained", "lines", "survivors"]


class Pipeline:
    """processor_split_parse_filter_gpu (include/lc_processor.h lc_pipeline_*): the reference's benchmark pipeline
    split -> processor_parse_regex_native -> processor_filter_regex_native, one device trip per read buffer when it can be fused.
    config = {"Split": {...}, "Parse": {...}, "Filter": {...}, "Fused": True}"""

    def __init__(self, config):
        self._L = _lib()
        h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(512)
        rc = self._L.lc_pipeline_create(json.dumps(config).encode("utf-8"), ctypes.byref(h), err, 512)
        if rc != binding.LC_OK:
            raise ProcessorInitError(err.value.decode("utf-8", "replace"))
        self._h = h

    @property
    def fused(self):
        return bool(self._L.lc_pipeline_is_fused(self._h))

    def process(self, group: EventGroup):
        rc = self._L.lc_pipeline_process(self._h, group._h)
        if rc == binding.LC_ERR_NO_DEVICE:
            raise binding.GpuUnavailableError("lc_pipeline_process: no usable HIP device (the pipeline has no CPU path)")
        if rc != binding.LC_OK:
            raise RuntimeError("lc_pipeline_process rc=%d" % rc)
        return group

    def counters(self):
        parse = (ctypes.c_uint64 * len(COUNTER_NAMES))()
        pipe = (ctypes.c_uint64 * len(PIPE_COUNTER_NAMES))()
        self._L.lc_pipeline_counters(self._h, parse, pipe)
        out = dict(zip(COUNTER_NAMES, [int(x) for x in parse]))
        out.update(zip(PIPE_COUNTER_NAMES, [int(x) for x in pipe]))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._L.lc_pipeline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def none_utf8(data: bytes):
    """ProcessorFilterNative::noneUtf8 as compiled into the library -> (is_bad, blanked copy)"""
    buf = ctypes.create_string_buffer(data, len(data))
    bad = _lib().lc_filter_none_utf8(buf, len(data), 0)    # CheckNoneUtf8
    _lib().lc_filter_none_utf8(buf, len(data), 1)          # FilterNoneUtf8
    return bool(bad), buf.raw[:len(data)]


class Filter:
    """processor_filter_regex_gpu; same config keys as processor_filter_regex_native (ConditionExp | FilterKey+FilterRegex |
    Include, DiscardingNonUTF8)."""

    MODES = ("bypass", "expression", "rule")

    def __init__(self, config):
        text = config if isinstance(config, str) else json.dumps(config)
        self._L = _lib()
        h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(512)
        rc = self._L.lc_filter_create(text.encode("utf-8"), ctypes.byref(h), err, 512)
        if rc != binding.LC_OK:
            raise ProcessorInitError(err.value.decode())
        self._h = h

    @property
    def mode(self):
        return self.MODES[self._L.lc_filter_mode(self._h)]

    def process(self, group: EventGroup):
        rc = self._L.lc_filter_process(self._h, self._L.lc_group_native(group._h))
        if rc == binding.LC_ERR_NO_DEVICE:
            raise binding.GpuUnavailableError("processor_filter_regex_gpu: no usable HIP device (no CPU path)")
        if rc != binding.LC_OK:
            raise RuntimeError("lc_filter_process rc=%d" % rc)

    def counters(self):
        buf = (ctypes.c_uint64 * 2)()
        self._L.lc_filter_counters(self._h, buf)
        return {"in_events_total": int(buf[0]), "out_events_total": int(buf[1])}

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.lc_filter_destroy(self._h)
            self._h = None
