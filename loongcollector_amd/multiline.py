"""ctypes binding of the multiline splitter (include/lc_multiline.h); plumbing for tests and tools."""
import ctypes
import json

from . import binding


class MultilineInitError(ValueError):
    pass


class _Record(ctypes.Structure):
    _fields_ = [("begin", ctypes.c_uint32), ("length", ctypes.c_uint32), ("matched", ctypes.c_uint32)]


class Multiline:
    def __init__(self, **config):
        L = self._L = binding.load()
        if not getattr(L, "_lc_multiline_bound", False):
            vp, cp, sz = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t
            L.lc_multiline_create.restype = ctypes.c_int
            L.lc_multiline_create.argtypes = [cp, sz, ctypes.POINTER(vp), cp, sz]
            L.lc_multiline_free.argtypes = [vp]
            L.lc_multiline_is_multiline.argtypes = [vp]
            L.lc_multiline_patterns.argtypes = [vp]
            L.lc_multiline_warnings.restype = cp
            L.lc_multiline_warnings.argtypes = [vp]
            L.lc_multiline_split_host.restype = ctypes.c_int
            L.lc_multiline_split_host.argtypes = [vp, cp, ctypes.c_uint32, ctypes.POINTER(ctypes.POINTER(_Record)),
                                                  ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
            L.lc_multiline_free_records.argtypes = [ctypes.POINTER(_Record)]
            L.lc_multiline_process_group.restype = ctypes.c_int
            L.lc_multiline_process_group.argtypes = [vp, vp]
            L.lc_multiline_counters.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
            L.lc_merge_multiline_create.restype = ctypes.c_int
            L.lc_merge_multiline_create.argtypes = [cp, sz, ctypes.POINTER(vp), cp, sz]
            L.lc_merge_multiline_free.argtypes = [vp]
            L.lc_merge_multiline_process_group.restype = ctypes.c_int
            L.lc_merge_multiline_process_group.argtypes = [vp, vp]
            L.lc_merge_multiline_counters.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
            L._lc_multiline_bound = True
        text = json.dumps(config).encode("utf-8")
        h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(512)
        if L.lc_multiline_create(text, len(text), ctypes.byref(h), err, 512) != 0:
            raise MultilineInitError(err.value.decode("utf-8", "replace"))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.lc_multiline_free(self._h)
            self._h = None

    @property
    def is_multiline(self):
        return bool(self._L.lc_multiline_is_multiline(self._h))

    @property
    def warnings(self):
        return self._L.lc_multiline_warnings(self._h).decode("utf-8", "replace")

    @property
    def patterns(self):
        m = self._L.lc_multiline_patterns(self._h)
        return {"start": bool(m & 1), "continue": bool(m & 2), "end": bool(m & 4)}

    def split(self, value: bytes):
        """-> (records [(begin, length, matched)], (input lines, unmatched lines, matched logs))"""
        recs = ctypes.POINTER(_Record)()
        n = ctypes.c_uint32()
        counters = (ctypes.c_uint32 * 3)()
        rc = self._L.lc_multiline_split_host(self._h, value, len(value), ctypes.byref(recs), ctypes.byref(n), counters)
        binding._check(rc, "lc_multiline_split_host")
        try:
            return [(recs[i].begin, recs[i].length, recs[i].matched & 1) for i in range(n.value)], tuple(counters)
        finally:
            self._L.lc_multiline_free_records(recs)


    def split_count(self, value: bytes):
        """the same call without building Python objects for the records (throughput measurements: the list of tuples costs more
        than the device trip) -> (number of records, counters)"""
        recs = ctypes.POINTER(_Record)()
        n = ctypes.c_uint32()
        counters = (ctypes.c_uint32 * 3)()
        rc = self._L.lc_multiline_split_host(self._h, value, len(value), ctypes.byref(recs), ctypes.byref(n), counters)
        binding._check(rc, "lc_multiline_split_host")
        self._L.lc_multiline_free_records(recs)
        return n.value, tuple(counters)

    def split_raw(self, value: bytes):
        """-> records [(begin, length, flag word)] with the LC_ML_LAST / LC_ML_RUN bits"""
        recs = ctypes.POINTER(_Record)()
        n = ctypes.c_uint32()
        counters = (ctypes.c_uint32 * 3)()
        rc = self._L.lc_multiline_split_host(self._h, value, len(value), ctypes.byref(recs), ctypes.byref(n), counters)
        binding._check(rc, "lc_multiline_split_host")
        try:
            return [(recs[i].begin, recs[i].length, recs[i].matched) for i in range(n.value)]
        finally:
            self._L.lc_multiline_free_records(recs)

    def process(self, group):
        """ProcessorSplitMultilineLogStringNative::Process on an EventGroup (loongcollector_amd.processor.EventGroup)."""
        from .processor import _lib
        binding._check(self._L.lc_multiline_process_group(self._h, _lib().lc_group_native(group._h)), "lc_multiline_process_group")

    def counters(self):
        """-> (matched lines, unmatched lines, matched events)"""
        c = (ctypes.c_uint64 * 3)()
        self._L.lc_multiline_counters(self._h, c)
        return tuple(int(x) for x in c)


class MergeMultiline:
    """processor_merge_multiline_log_native (MergeType "regex" or "flag") on event groups."""

    def __init__(self, **config):
        Multiline.__new__(Multiline)  # (binds the library's prototypes once)
        try:
            Multiline(StartPattern="x")
        except Exception:
            pass
        L = self._L = binding.load()
        text = json.dumps(config).encode("utf-8")
        h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(512)
        if L.lc_merge_multiline_create(text, len(text), ctypes.byref(h), err, 512) != 0:
            raise MultilineInitError(err.value.decode("utf-8", "replace"))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.lc_merge_multiline_free(self._h)
            self._h = None

    def process(self, group):
        from .processor import _lib
        binding._check(self._L.lc_merge_multiline_process_group(self._h, _lib().lc_group_native(group._h)),
                       "lc_merge_multiline_process_group")

    def patterns(self):
        """bit 0 start, bit 1 continue, bit 2 end: the patterns the processor matches with -- MultilineOptions' own regexes (a trailing
        '$' and ".*"s stripped, ContinuePattern dropped when all three are given), not the splitter's reading; 0 in flag mode"""
        self._L.lc_merge_multiline_patterns.argtypes = [ctypes.c_void_p]
        return int(self._L.lc_merge_multiline_patterns(self._h))

    def warnings(self):
        self._L.lc_merge_multiline_warnings.restype = ctypes.c_char_p
        self._L.lc_merge_multiline_warnings.argtypes = [ctypes.c_void_p]
        return self._L.lc_merge_multiline_warnings(self._h).decode("utf-8", "replace")

    def counters(self):
        """-> (merged events, unmatched events)"""
        c = (ctypes.c_uint64 * 2)()
        self._L.lc_merge_multiline_counters(self._h, c)
        return tuple(int(x) for x in c)


ML_HAS_START, ML_HAS_CONT, ML_HAS_END, ML_DISCARD, ML_FLUSH = 1, 2, 4, 8, 16
ML_LAST, ML_RUN = 0x80000000, 2


def bounds_model(mode, flags, off=None, nbytes=0):
    """lc_multiline_bounds_model: the device scan's code run on the host, slice by slice.  flags: one byte per item (bit 0 start,
    bit 1 continue, bit 2 end matched); off: the offsets[n+1] + separator table (byte records) or None (item records).
    -> (records [(begin, length, flags)], counts[8])"""
    import numpy as np
    L = binding.load()
    L.lc_multiline_bounds_model.restype = ctypes.c_int
    L.lc_multiline_bounds_model.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32,
                                            ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    fl = np.ascontiguousarray(np.asarray(flags, dtype=np.uint8))
    n = int(fl.size)
    cap = n + 2
    recs = np.zeros((cap, 3), dtype=np.uint32)
    counts = np.zeros(8, dtype=np.uint32)
    o = None if off is None else np.ascontiguousarray(np.asarray(off, dtype=np.uint32))
    rc = L.lc_multiline_bounds_model(mode, fl.ctypes.data if n else None, n, None if o is None else o.ctypes.data, nbytes,
                                     recs.ctypes.data, cap, counts.ctypes.data)
    binding._check(rc, "lc_multiline_bounds_model")
    return [tuple(int(x) for x in r) for r in recs[:int(counts[3])]], [int(x) for x in counts]
