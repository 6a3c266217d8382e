"""ctypes binding of the Grok processor (include/lc_grok.h).  Plumbing for tests, tools and bench.py only: the product
is the C ABI; matching always runs on the HIP device (there is no CPU path)."""
import ctypes
import json

import numpy as np

from . import binding


class GrokInitError(ValueError):
    pass


def _lib():
    L = binding.load()
    if not getattr(L, "_lc_grok_bound", False):
        vp, cp, sz, i32, u32 = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32
        L.lc_grok_create.restype = i32
        L.lc_grok_create.argtypes = [cp, sz, ctypes.POINTER(vp), cp, sz]
        L.lc_grok_free.argtypes = [vp]
        L.lc_grok_wait_ready.restype = None
        L.lc_grok_wait_ready.argtypes = [vp]
        for name in ("lc_grok_match_count", "lc_grok_key_count", "lc_grok_row_ints"):
            getattr(L, name).restype = i32
            getattr(L, name).argtypes = [vp]
        L.lc_grok_expanded.restype = cp
        L.lc_grok_expanded.argtypes = [vp, i32]
        L.lc_grok_processed.restype = cp
        L.lc_grok_processed.argtypes = [vp, cp]
        L.lc_grok_denormalize.restype = vp
        L.lc_grok_denormalize.argtypes = [vp, cp, cp, sz]
        L.lc_grok_engine.restype = i32
        L.lc_grok_engine.argtypes = [vp, i32]
        L.lc_grok_key.restype = cp
        L.lc_grok_key.argtypes = [vp, i32]
        L.lc_grok_column_count.restype = i32
        L.lc_grok_column_count.argtypes = [vp, i32]
        L.lc_grok_column_key.restype = i32
        L.lc_grok_column_key.argtypes = [vp, i32, i32]
        L.lc_grok_scratch_bytes.restype = sz
        L.lc_grok_scratch_bytes.argtypes = [vp, u32]
        L.lc_grok_match_device.restype = i32
        L.lc_grok_match_device.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, u32, vp, vp, sz, vp]
        L.lc_grok_match_host.restype = i32
        L.lc_grok_match_host.argtypes = [vp, vp, vp, vp, u32, vp, ctypes.POINTER(vp)]
        L.lc_grok_result_arrays.argtypes = [vp] + [ctypes.POINTER(vp)] * 4
        L.lc_grok_result_free.argtypes = [vp]
        L.lc_grok_process_logs_json.restype = i32
        L.lc_grok_process_logs_json.argtypes = [vp, cp, sz, ctypes.POINTER(vp)]
        L.lc_grok_free_string.argtypes = [vp]
        L.lc_grok_last_batch_stats.restype = None
        L.lc_grok_last_batch_stats.argtypes = [vp]
        L._lc_grok_bound = True
    return L


class Grok:
    """processor_grok with the reference's config keys: Grok(Match=[...], CustomPatterns={...}, KeepSource=False, ...)"""

    def __init__(self, **config):
        self._L = _lib()
        text = json.dumps(config).encode("utf-8")
        h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(1024)
        rc = self._L.lc_grok_create(text, len(text), ctypes.byref(h), err, 1024)
        if rc != 0:
            raise GrokInitError(err.value.decode("utf-8", "replace"))
        self._h = h
        self.n_match = self._L.lc_grok_match_count(h)
        self.keys = [self._L.lc_grok_key(h, k).decode("utf-8") for k in range(self._L.lc_grok_key_count(h))]
        self.row_ints = self._L.lc_grok_row_ints(h)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.lc_grok_free(self._h)
            self._h = None

    def literal_index(self):
        """lc_grok_literal_index: the Aho-Corasick DFA over the required literals of the Match list (uint32 words), or None"""
        import numpy as np
        words = ctypes.c_void_p()
        n = ctypes.c_size_t()
        self._L.lc_grok_literal_index.restype = ctypes.c_int
        self._L.lc_grok_literal_index.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        rc = self._L.lc_grok_literal_index(self._h, ctypes.byref(words), ctypes.byref(n))
        if rc != 0 or not words.value:
            return None
        return np.ctypeslib.as_array(ctypes.cast(words.value, ctypes.POINTER(ctypes.c_uint32)), shape=(n.value,)).copy()

    def wait_ready(self):
        """lc_grok_wait_ready: block until the warm-up thread has compiled the anchored searches (speed only, never results)"""
        self._L.lc_grok_wait_ready(self._h)
        return self

    def expanded(self, i):
        return self._L.lc_grok_expanded(self._h, i).decode("utf-8")

    def processed(self, name):
        p = self._L.lc_grok_processed(self._h, name.encode("utf-8"))
        return None if p is None else p.decode("utf-8")

    def denormalize(self, pattern):
        """expand %{...} references against this handle's library without compiling the result"""
        err = ctypes.create_string_buffer(512)
        p = self._L.lc_grok_denormalize(self._h, pattern.encode("utf-8"), err, 512)
        if not p:
            raise GrokInitError(err.value.decode("utf-8", "replace"))
        try:
            return ctypes.string_at(p).decode("utf-8")
        finally:
            self._L.lc_grok_free_string(p)

    def engine(self, i):
        return self._L.lc_grok_engine(self._h, i)

    def columns(self, i):
        """emitted key (or None for a numbered group) of every capture column of Match[i]"""
        out = []
        for c in range(self._L.lc_grok_column_count(self._h, i)):
            k = self._L.lc_grok_column_key(self._h, i, c)
            out.append(None if k < 0 else self.keys[k])
        return out

    # ---- processGrok over a batch of values (bytes); -> (pattern int32[n], [[(key, value bytes), ...] per value])
    def match_host(self, values):
        n = len(values)
        data = np.frombuffer(b"".join(values) + b"\0" * 16, dtype=np.uint8)
        length = np.array([len(v) for v in values], dtype=np.uint32)
        off = np.zeros(n, dtype=np.uint32)
        if n:
            off[1:] = np.cumsum(length[:-1], dtype=np.uint64).astype(np.uint32)
        pattern = np.empty(n, dtype=np.int32)
        res = ctypes.c_void_p()
        rc = self._L.lc_grok_match_host(self._h, data.ctypes.data, off.ctypes.data, length.ctypes.data, n,
                                        pattern.ctypes.data, ctypes.byref(res))
        binding._check(rc, "lc_grok_match_host")
        try:
            ptrs = [ctypes.c_void_p() for _ in range(4)]
            self._L.lc_grok_result_arrays(res, *[ctypes.byref(p) for p in ptrs])
            foff = np.ctypeslib.as_array(ctypes.cast(ptrs[0], ctypes.POINTER(ctypes.c_uint32)), shape=(n + 1,)).copy()
            m = int(foff[n])
            key, beg, end = [np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint32)), shape=(m,)).copy()
                             if m else np.zeros(0, np.uint32) for p in ptrs[1:]]
        finally:
            self._L.lc_grok_result_free(res)
        fields = [[(self.keys[int(key[f])], values[i][int(beg[f]):int(end[f])]) for f in range(int(foff[i]), int(foff[i + 1]))]
                  for i in range(n)]
        return pattern, fields

    # ---- device-resident batch: torch tensors in, torch tensors out (bench.py, tests)
    def scratch_bytes(self, n):
        return self._L.lc_grok_scratch_bytes(self._h, n)

    def match_device(self, d_data, d_off, d_len, n, d_pattern, d_first, d_extra, d_nextra, d_scratch, stream=None):
        rc = self._L.lc_grok_match_device(self._h, d_data.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), n,
                                          d_pattern.data_ptr(), d_first.data_ptr(), d_extra.data_ptr(),
                                          d_extra.shape[0], d_nextra.data_ptr(), d_scratch.data_ptr(),
                                          d_scratch.numel() * d_scratch.element_size(), stream)
        binding._check(rc, "lc_grok_match_device")

    def lazy_settle(self, timeout_ms=60000):
        """block until the lazy automata's trainer has nothing to do (include/lc_grok.h); -> True if it settled in time"""
        self._L.lc_grok_lazy_settle.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        return self._L.lc_grok_lazy_settle(self._h, timeout_ms) == 0

    def lazy_stats(self):
        w = (ctypes.c_uint64 * 5)()
        self._L.lc_grok_lazy_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self._L.lc_grok_lazy_stats(self._h, ctypes.cast(w, ctypes.c_void_p))
        return dict(zip(("automata_in_use", "builds", "values_offered", "values_kept", "batches_taken"), (int(x) for x in w)))

    def combiner_stats(self):
        """the group commit behind match_host (csrc/group_combiner.hpp) since the handle was created"""
        w = (ctypes.c_uint64 * 11)()
        self._L.lc_grok_combiner_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self._L.lc_grok_combiner_stats(self._h, ctypes.cast(w, ctypes.c_void_p))
        out = {"batches": int(w[0]), "groups": int(w[1]), "values": int(w[2]), "largest_batch_groups": int(w[3]), "linger_expired": int(w[4])}
        out["worker_us"] = dict(zip(("idle", "linger", "place", "gather", "run", "take_out"), (int(x) for x in w[5:11])))
        return out

    def last_batch_stats(self):
        """what the calling thread's last device batch did: host syncs, active entries, (entry, value) pairs, deferred entries, path"""
        w = (ctypes.c_uint32 * 5)()
        self._L.lc_grok_last_batch_stats(ctypes.cast(w, ctypes.c_void_p))
        return {"host_syncs": int(w[0]), "active_entries": int(w[1]), "pairs": int(w[2]), "deferred_entries": int(w[3]),
                "speculative": bool(w[4])}

    # ---- ProcessLogs: logs = [[(key, value str), ...], ...] -> same shape
    def process_logs(self, logs):
        text = json.dumps([[list(kv) for kv in log] for log in logs]).encode("utf-8")
        out = ctypes.c_void_p()
        rc = self._L.lc_grok_process_logs_json(self._h, text, len(text), ctypes.byref(out))
        binding._check(rc, "lc_grok_process_logs_json")
        try:
            return [[tuple(kv) for kv in log] for log in json.loads(ctypes.string_at(out).decode("utf-8"))]
        finally:
            self._L.lc_grok_free_string(out)


class GoRegex:
    """Go plugin processor_regex (include/lc_go_regex.h): GoRegex(Regex=..., Keys=[...], FullMatch=False, ...)"""

    def __init__(self, **config):
        L = self._L = binding.load()
        if not getattr(L, "_lc_goregex_bound", False):
            vp, cp, sz = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t
            L.lc_goregex_create.restype = ctypes.c_int
            L.lc_goregex_create.argtypes = [cp, sz, ctypes.POINTER(vp), cp, sz]
            L.lc_goregex_free.argtypes = [vp]
            L.lc_goregex_process_logs_json.restype = ctypes.c_int
            L.lc_goregex_process_logs_json.argtypes = [vp, cp, sz, ctypes.POINTER(vp)]
            L.lc_goregex_free_string.argtypes = [vp]
            L._lc_goregex_bound = True
        text = json.dumps(config).encode("utf-8")
        h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(512)
        if L.lc_goregex_create(text, len(text), ctypes.byref(h), err, 512) != 0:
            raise GrokInitError(err.value.decode("utf-8", "replace"))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.lc_goregex_free(self._h)
            self._h = None

    def process_logs(self, logs):
        text = json.dumps([[list(kv) for kv in log] for log in logs]).encode("utf-8")
        out = ctypes.c_void_p()
        rc = self._L.lc_goregex_process_logs_json(self._h, text, len(text), ctypes.byref(out))
        binding._check(rc, "lc_goregex_process_logs_json")
        try:
            return [[tuple(kv) for kv in log] for log in json.loads(ctypes.string_at(out).decode("utf-8"))]
        finally:
            self._L.lc_goregex_free_string(out)
