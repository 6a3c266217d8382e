"""ctypes binding of liblc_regex_gpu.so (the C ABI declared in include/lc_regex_gpu.h).

The library is the product; this module is only the Python-side plumbing used by tests and bench.py
(device memory comes from torch tensors, whose data_ptr() values are passed straight through).
There is no CPU execution path: every match call needs a HIP device and raises otherwise.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LC_REGEX_GPU_LIB", os.path.join(_HERE, "lib", "liblc_regex_gpu.so"))

LC_ENGINE_AUTO, LC_ENGINE_TDFA, LC_ENGINE_NFA, LC_ENGINE_DECIDE, LC_ENGINE_BT = 0, 1, 2, 3, 4
LC_NOMATCH, LC_MATCH, LC_OVERFLOW, LC_GAVE_UP = 0, 1, 2, 3
LC_OK, LC_ERR_SYNTAX, LC_ERR_UNSUPPORTED, LC_ERR_NO_DEVICE, LC_ERR_HIP, LC_ERR_ARG = range(6)
LC_SYNTAX_ICASE, LC_SYNTAX_NO_DOTALL, LC_SYNTAX_NO_MULTILINE, LC_SYNTAX_EXTENDED, LC_SYNTAX_NAMED_ONLY = 1, 2, 4, 8, 16
LC_SYNTAX_SEARCH = 32
LC_SYNTAX_REGEXP2 = 64
LC_SYNTAX_PREFIX = 128

(LC_TABLE_CLASSMAP, LC_TABLE_TDFA_TRANS, LC_TABLE_TDFA_OPSSTART, LC_TABLE_TDFA_OPS, LC_TABLE_TDFA_FINALID,
 LC_TABLE_TDFA_FINALMAP, LC_TABLE_TDFA_HEADER, LC_TABLE_NFA_BLOB, LC_TABLE_TDFA_STARTAFTER, LC_TABLE_TDFA_BLOB,
 LC_TABLE_TDFA_WIDE_BLOB, LC_TABLE_TDFA_L2_BLOB) = range(12)
LC_TABLE_LAZY_TDFA_BLOB = 12
LC_TABLE_BT_BLOB = 13


class LcRegexInfo(ctypes.Structure):
    _fields_ = [("engine", ctypes.c_int), ("mark_count", ctypes.c_int), ("positions", ctypes.c_uint32),
                ("states", ctypes.c_uint32), ("classes", ctypes.c_uint32), ("registers", ctypes.c_uint32),
                ("table_bytes", ctypes.c_uint32)]


class RegexSyntaxError(ValueError):
    pass


class RegexUnsupportedError(ValueError):
    pass


class GpuUnavailableError(RuntimeError):
    pass


_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so (same SONAME as /opt/rocm's) and dlopens it
    by path; this library is linked against the SONAME.  Loaded after torch it binds to torch's copy; loaded BEFORE torch
    (build() then smoke() in one process) the process would end up with two runtimes, and the second one to initialise
    finds no device.  So torch's copy is mapped first whenever torch is installed -- without importing torch."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        hip = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(hip):
            ctypes.CDLL(hip, mode=ctypes.RTLD_GLOBAL)
    except OSError:
        pass


def load(path=None):
    """Load the shared library (raises OSError if it has not been built: run __graft_entry__.build())."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    _share_torch_hip_runtime()
    L = ctypes.CDLL(path or LIB_PATH)
    vp, cp, sz, u32, i32 = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int
    L.lc_regex_compile.restype = i32
    L.lc_regex_compile.argtypes = [cp, sz, u32, i32, ctypes.POINTER(vp), cp, sz]
    L.lc_regex_free.argtypes = [vp]
    L.lc_regex_mark_count.argtypes = [vp]
    L.lc_regex_group_name.restype = cp
    L.lc_regex_group_name.argtypes = [vp, i32]
    L.lc_regex_info.argtypes = [vp, ctypes.POINTER(LcRegexInfo)]
    L.lc_regex_table.argtypes = [vp, i32, ctypes.POINTER(vp), ctypes.POINTER(sz)]
    L.lc_device_count.restype = i32
    L.lc_regex_match_device.restype = i32
    L.lc_regex_match_device.argtypes = [vp, vp, vp, vp, u32, u32, u32, vp, vp, vp]
    L.lc_regex_match_device_engine.restype = i32
    L.lc_regex_match_device_engine.argtypes = [vp, i32, vp, vp, vp, u32, u32, u32, vp, vp, vp]
    L.lc_regex_match_device_dyn.restype = i32
    L.lc_regex_match_device_dyn.argtypes = [vp, i32, vp, vp, u32, vp, u32, u32, vp, vp, vp]
    L.lc_sched_scratch_bytes.restype = sz
    L.lc_sched_scratch_bytes.argtypes = [u32]
    L.lc_regex_match_device_ragged.restype = i32
    L.lc_regex_match_device_ragged.argtypes = [vp, i32, vp, vp, vp, u32, u32, vp, u32, vp, vp, vp, sz, vp]
    L.lc_regex_compile_screen.restype = vp
    L.lc_regex_compile_screen.argtypes = [ctypes.c_char_p, sz, u32, u32, sz]
    L.lc_regex_compile_relaxed_screen.restype = vp
    L.lc_regex_compile_relaxed_screen.argtypes = [ctypes.c_char_p, sz, u32, u32, sz]
    L.lc_regex_screen_device.restype = i32
    L.lc_regex_screen_device.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, vp]
    L.lc_regex_required_literal.restype = vp
    L.lc_regex_required_literal.argtypes = [vp, ctypes.POINTER(sz)]
    L.lc_regex_run_captures.restype = i32
    L.lc_regex_run_captures.argtypes = [vp, vp, vp, i32]
    L.lc_regex_match_device_from.restype = i32
    L.lc_regex_match_device_from.argtypes = [vp, i32, vp, vp, vp, u32, u32, vp, vp, vp, u32, vp, vp, vp]
    L.lc_split_scratch_bytes.restype = sz
    L.lc_split_scratch_bytes.argtypes = [ctypes.c_uint64]
    L.lc_split_lines_device.restype = i32
    L.lc_split_lines_device.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint8, vp, u32, vp, vp, sz, vp]
    L.lc_regex_match_host.restype = i32
    L.lc_regex_match_host.argtypes = [vp, vp, vp, vp, u32, u32, vp, vp]
    L.lc_last_error.restype = cp
    if path is None:
        _lib = L
    return L


def _check(rc, what):
    if rc == LC_OK:
        return
    if rc == LC_ERR_NO_DEVICE:
        raise GpuUnavailableError("%s: no usable HIP device (the engine has no CPU path)" % what)
    msg = load().lc_last_error()
    raise RuntimeError("%s failed: rc=%d %s" % (what, rc, msg.decode() if msg else ""))


class LcMatchJob(ctypes.Structure):
    _fields_ = [("re", ctypes.c_void_p), ("d_data", ctypes.c_void_p), ("d_off", ctypes.c_void_p), ("d_len", ctypes.c_void_p),
                ("sep_bytes", ctypes.c_uint32), ("n", ctypes.c_uint32), ("ngroups", ctypes.c_uint32),
                ("d_caps", ctypes.c_void_p), ("d_status", ctypes.c_void_p)]


def make_jobs(jobs):
    """jobs: list of (GpuRegex, d_data, d_off, d_len or None, n, d_caps, d_status, sep_bytes) with torch device tensors
    -> a marshalled job array for match_device_multi (keep it alive until the stream has been synchronised)."""
    arr = (LcMatchJob * len(jobs))()
    for k, (rx, d_data, d_off, d_len, n, d_caps, d_status, sep) in enumerate(jobs):
        arr[k] = LcMatchJob(rx.handle, d_data.data_ptr(), d_off.data_ptr(), d_len.data_ptr() if d_len is not None else None,
                            sep, n, rx.groups, d_caps.data_ptr(), d_status.data_ptr())
    return arr


def match_device_multi(job_array, stream=None):
    """lc_regex_match_device_multi: several batches, each with its own regex, packed into one kernel launch."""
    L = load()
    L.lc_regex_match_device_multi.restype = ctypes.c_int
    L.lc_regex_match_device_multi.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    _check(L.lc_regex_match_device_multi(ctypes.cast(job_array, ctypes.c_void_p), len(job_array), ctypes.c_void_p(stream or 0)),
           "lc_regex_match_device_multi")


def launched_kernels():
    """Names of the match kernels this thread has launched since the last call (lc_launched_kernels)."""
    L = load()
    L.lc_launched_kernels.restype = ctypes.c_size_t
    L.lc_launched_kernels.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    buf = ctypes.create_string_buffer(1024)
    L.lc_launched_kernels(buf, 1024)
    return buf.value.decode()


class GpuRegex:
    """Compiled pattern handle; mirrors the role of `boost::regex` in ProcessorParseRegexNative (mReg)."""

    def __init__(self, pattern, syntax_flags=0, engine=LC_ENGINE_AUTO, lib=None):
        self._L = lib or load()
        if isinstance(pattern, str):
            pattern = pattern.encode("utf-8")
        h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(512)
        rc = self._L.lc_regex_compile(pattern, len(pattern), syntax_flags, engine, ctypes.byref(h), err, 512)
        if rc == LC_ERR_SYNTAX:
            raise RegexSyntaxError(err.value.decode())
        if rc != LC_OK:
            raise RegexUnsupportedError(err.value.decode())
        self._h = h
        self.pattern = pattern
        self.groups = self._L.lc_regex_mark_count(h)

    @classmethod
    def compile_screen(cls, pattern, syntax_flags=0, max_states=1024, max_table_bytes=32 * 1024, lib=None, relaxed=False):
        """lc_regex_compile_screen: TDFA handle for the longest affordable prefix of `pattern`, or None;
        relaxed=True: lc_regex_compile_relaxed_screen, the whole pattern relaxed until its automaton is small"""
        L = lib or load()
        if isinstance(pattern, str):
            pattern = pattern.encode("utf-8")
        fn = L.lc_regex_compile_relaxed_screen if relaxed else L.lc_regex_compile_screen
        h = fn(pattern, len(pattern), syntax_flags, max_states, max_table_bytes)
        if not h:
            return None
        self = cls.__new__(cls)
        self._L = L
        self._h = ctypes.c_void_p(h)
        self.pattern = pattern
        self.groups = L.lc_regex_mark_count(self._h)
        return self

    def close(self):
        if getattr(self, "_h", None):
            self._L.lc_regex_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def info(self):
        i = LcRegexInfo()
        _check(self._L.lc_regex_info(self._h, ctypes.byref(i)), "lc_regex_info")
        return {f: getattr(i, f) for f, _ in LcRegexInfo._fields_}

    def atomic_groups(self):
        """-> (kept, elided): atomic group instances the engines honour / groups made plain at compile time (atomic_elide.cpp)"""
        k, e = ctypes.c_uint32(), ctypes.c_uint32()
        self._L.lc_regex_atomic_groups.restype = None
        self._L.lc_regex_atomic_groups.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
        self._L.lc_regex_atomic_groups(self._h, ctypes.byref(k), ctypes.byref(e))
        return int(k.value), int(e.value)

    def has_nfa_program(self):
        """False when the follow NFA does not fit the NFA kernel's format (then only the TDFA engine can run it)."""
        p = ctypes.c_void_p()
        n = ctypes.c_size_t()
        return self._L.lc_regex_table(self._h, LC_TABLE_NFA_BLOB, ctypes.byref(p), ctypes.byref(n)) == LC_OK

    def group_name(self, g):
        r = self._L.lc_regex_group_name(self._h, g)
        return r.decode() if r else None

    def lazy_train(self, values):
        """lc_regex_lazy_train: adds `values` (bytes objects) to the handle's sample and (re)builds its partial tagged DFA along them.
        -> {"states", "transitions", "sample", "sample_misses", "in_use"}"""
        n = len(values)
        length = np.array([len(v) for v in values], dtype=np.uint32)
        off = np.zeros(max(n, 1), dtype=np.uint32)
        if n > 1:
            off[1:n] = np.cumsum(length[:-1], dtype=np.uint64).astype(np.uint32)
        data = np.frombuffer(b"".join(values) + b"\0" * 16, dtype=np.uint8).copy()
        out = (ctypes.c_uint64 * 5)()
        self._L.lc_regex_lazy_train.restype = ctypes.c_int
        self._L.lc_regex_lazy_train.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_uint32, ctypes.c_void_p]
        _check(self._L.lc_regex_lazy_train(self._h, data.ctypes.data, off.ctypes.data, length.ctypes.data, n, ctypes.cast(out, ctypes.c_void_p)),
               "lc_regex_lazy_train")
        return dict(zip(("states", "transitions", "sample", "sample_misses", "in_use"), (int(x) for x in out)))

    def table(self, which, dtype):
        p = ctypes.c_void_p()
        n = ctypes.c_size_t()
        rc = self._L.lc_regex_table(self._h, which, ctypes.byref(p), ctypes.byref(n))
        if rc != LC_OK:
            return None
        if n.value == 0 or not p.value:
            return np.zeros((0,), dtype=dtype)
        buf = (ctypes.c_uint8 * n.value).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype).copy()

    # ---- device-resident batch (torch tensors on the current HIP device)
    def match_device(self, d_data, d_off, d_len, n, d_caps, d_status, ngroups=None, sep_bytes=0, stream=None,
                     engine=LC_ENGINE_AUTO):
        G = self.groups if ngroups is None else ngroups
        rc = self._L.lc_regex_match_device_engine(self._h, engine, d_data.data_ptr(), d_off.data_ptr(),
                                                  d_len.data_ptr() if d_len is not None else None, sep_bytes, n, G,
                                                  d_caps.data_ptr(), d_status.data_ptr(), stream)
        _check(rc, "lc_regex_match_device")

    def match_device_ragged(self, d_data, d_off, d_len, n, d_caps, d_status, d_scratch, ngroups=None, sep_bytes=0,
                            d_nlines=None, stream=None, engine=LC_ENGINE_AUTO):
        """length-scheduled match for ragged batches; d_scratch: >= sched_scratch_bytes(n) bytes on the device"""
        G = self.groups if ngroups is None else ngroups
        rc = self._L.lc_regex_match_device_ragged(
            self._h, engine, d_data.data_ptr(), d_off.data_ptr(), d_len.data_ptr() if d_len is not None else None,
            sep_bytes, n, d_nlines.data_ptr() if d_nlines is not None else None, G, d_caps.data_ptr(),
            d_status.data_ptr(), d_scratch.data_ptr(), d_scratch.numel() * d_scratch.element_size(), stream)
        _check(rc, "lc_regex_match_device_ragged")

    def prefer_wave_tdfa(self):
        """lc_regex_prefer_wave_tdfa: small batches of this handle take tdfa_wave_kernel (and the handle gets its global-memory tables)"""
        self._L.lc_regex_prefer_wave_tdfa.restype = ctypes.c_int
        self._L.lc_regex_prefer_wave_tdfa.argtypes = [ctypes.c_void_p]
        return bool(self._L.lc_regex_prefer_wave_tdfa(self._h))

    def required_literal(self):
        n = ctypes.c_size_t()
        p = self._L.lc_regex_required_literal(self._h, ctypes.byref(n))
        return ctypes.string_at(p, n.value) if n.value else b""

    def run_captures(self):
        """[(group index, frozenset of bytes)] of the groups written "(?=(S*))": the tables stamp their begin only"""
        n = self._L.lc_regex_run_captures(self._h, None, None, 0)
        if n == 0:
            return []
        groups = (ctypes.c_int32 * n)()
        sets = (ctypes.c_uint8 * (32 * n))()
        self._L.lc_regex_run_captures(self._h, groups, sets, n)
        return [(int(groups[i]), frozenset(b for b in range(256) if (sets[32 * i + b // 8] >> (b % 8)) & 1)) for i in range(n)]

    def match_device_from(self, d_data, d_off, d_len, n, d_caps, d_status, d_lines=None, d_nlines=None, d_from=None,
                          ngroups=None, sep_bytes=0, stream=None, engine=LC_ENGINE_AUTO):
        """subset of lines (d_lines) and/or searches resumed inside their line (d_from, indexed by line)"""
        G = self.groups if ngroups is None else ngroups
        ptr = lambda t: t.data_ptr() if t is not None else None
        rc = self._L.lc_regex_match_device_from(self._h, engine, d_data.data_ptr(), d_off.data_ptr(), ptr(d_len), sep_bytes,
                                                n, ptr(d_lines), ptr(d_nlines), ptr(d_from), G, d_caps.data_ptr(),
                                                d_status.data_ptr(), stream)
        _check(rc, "lc_regex_match_device_from")

    def screen_device(self, d_data, d_off, d_len, n, d_out, d_count, d_lines=None, stream=None):
        """lc_regex_screen_device: a relaxed screen's pass over the values; accepted values -> d_out, their number += d_count"""
        rc = self._L.lc_regex_screen_device(self._h, d_data.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), n,
                                            d_lines.data_ptr() if d_lines is not None else None, d_out.data_ptr(),
                                            d_count.data_ptr(), stream)
        _check(rc, "lc_regex_screen_device")

    def match_device_dyn(self, d_data, d_off, d_nlines, max_lines, d_caps, d_status, ngroups=None, sep_bytes=1,
                         stream=None, engine=LC_ENGINE_AUTO):
        G = self.groups if ngroups is None else ngroups
        rc = self._L.lc_regex_match_device_dyn(self._h, engine, d_data.data_ptr(), d_off.data_ptr(), sep_bytes,
                                               d_nlines.data_ptr(), max_lines, G, d_caps.data_ptr(),
                                               d_status.data_ptr(), stream)
        _check(rc, "lc_regex_match_device_dyn")

    # ---- host batch (numpy arrays); pinned double-buffered H2D/D2H inside the library
    def match_host(self, data, off, length, ngroups=None):
        G = self.groups if ngroups is None else ngroups
        data = np.ascontiguousarray(data, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        length = np.ascontiguousarray(length, dtype=np.uint32)
        n = int(off.shape[0])
        caps = np.empty((n, 2 * G), dtype=np.int32)
        status = np.empty((n,), dtype=np.uint8)
        rc = self._L.lc_regex_match_host(self._h, data.ctypes.data, off.ctypes.data, length.ctypes.data, n, G,
                                         caps.ctypes.data, status.ctypes.data)
        _check(rc, "lc_regex_match_host")
        return caps, status


def split_lines_device(d_data, nbytes, d_off, d_nlines, d_scratch, split_char=10, stream=None):
    """ProcessorSplitLogStringNative on the device; all arguments are torch tensors on the current HIP device."""
    L = load()
    rc = L.lc_split_lines_device(d_data.data_ptr(), nbytes, split_char, d_off.data_ptr(), d_off.numel(),
                                 d_nlines.data_ptr(), d_scratch.data_ptr(), d_scratch.numel() * d_scratch.element_size(),
                                 stream)
    _check(rc, "lc_split_lines_device")


def sched_scratch_bytes(n):
    return int(load().lc_sched_scratch_bytes(n))


def split_scratch_bytes(nbytes):
    return int(load().lc_split_scratch_bytes(nbytes))


def device_count():
    return load().lc_device_count()


# ---- thread -> device binding (include/lc_regex_gpu.h: lc_runtime_*bind*)
LC_BIND_INHERIT, LC_BIND_ROUND_ROBIN, LC_BIND_FIXED = 0, 1, 2


def _bind_lib():
    L = load()
    L.lc_runtime_set_bind_policy.restype = ctypes.c_int
    L.lc_runtime_set_bind_policy.argtypes = [ctypes.c_int, ctypes.c_int]
    L.lc_runtime_bind_policy.restype = ctypes.c_int
    L.lc_runtime_bind_thread.restype = ctypes.c_int
    L.lc_runtime_bind_thread.argtypes = [ctypes.c_int]
    L.lc_runtime_set_thread_device.restype = ctypes.c_int
    L.lc_runtime_set_thread_device.argtypes = [ctypes.c_int]
    L.lc_runtime_thread_device.restype = ctypes.c_int
    L.lc_runtime_device_for_ordinal.restype = ctypes.c_int
    L.lc_runtime_device_for_ordinal.argtypes = [ctypes.c_uint32, ctypes.c_int]
    return L


def set_bind_policy(policy, device=0):
    """Process-wide placement of the threads that enter the library through its HOST entry points (processors, match_host ...):
    LC_BIND_ROUND_ROBIN (library default: thread ordinal % visible devices), LC_BIND_FIXED (every thread -> `device`; what a
    process that owns ONE GPU -- a rank of bench.py -- says), LC_BIND_INHERIT (never switch)."""
    _check(_bind_lib().lc_runtime_set_bind_policy(policy, device), "lc_runtime_set_bind_policy")


def bind_policy():
    return _bind_lib().lc_runtime_bind_policy()


def bind_thread(policy=-1):
    """Bind the calling thread now; returns its device."""
    d = _bind_lib().lc_runtime_bind_thread(policy)
    if d < 0:
        _check(-d, "lc_runtime_bind_thread")
    return d


def set_thread_device(device):
    _check(_bind_lib().lc_runtime_set_thread_device(device), "lc_runtime_set_thread_device")


def thread_device():
    return _bind_lib().lc_runtime_thread_device()


def device_for_ordinal(ordinal, ndevices):
    return _bind_lib().lc_runtime_device_for_ordinal(ordinal, ndevices)
