"""oracle/oracle.py -- ctypes loader for the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (loongcollector_amd/) never does.  See oracle/README.md.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ORX_ICASE = 1
ORX_NO_MOD_S = 2
ORX_NO_MOD_M = 4
ORX_EXTENDED = 8
ORX_REGEXP2 = 16
ORX_NAMED_BACKREFS = 32


def build(force=False):
    """Compile liboracle.so with gcc (idempotent)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("bt_regex.c", "processor_oracle.c", "pcre_baseline.c", "grok_baseline.c", "bt_regex.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.orx_compile.restype = ctypes.c_void_p
        L.orx_compile.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_char_p, ctypes.c_size_t]
        L.orx_free.argtypes = [ctypes.c_void_p]
        L.orx_mark_count.argtypes = [ctypes.c_void_p]
        L.orx_group_name.restype = ctypes.c_char_p
        L.orx_group_name.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orx_fullmatch.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
        L.orx_search.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
        L.orx_prefixmatch.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
        L.orx_fullmatch_batch.restype = ctypes.c_long
        L.orx_fullmatch_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.orx_process_batch.restype = ctypes.c_ulong
        L.orx_process_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_ulong)]
        L.orx_pcre_version.restype = ctypes.c_char_p
        L.orx_pcre_fullmatch_batch.restype = ctypes.c_long
        L.orx_pcre_fullmatch_batch.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.orx_grok_first_match.restype = ctypes.c_long
        L.orx_grok_first_match.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        _LIB = L
    return _LIB


def pcre_version():
    """Version string of the PCRE1 library the baseline leg can open ("" if none)."""
    return lib().orx_pcre_version().decode()


def pcre_fullmatch_batch(pattern, data, off, length, ngroups, jit=False):
    """PCRE1 (BASELINE.md section 2's stand-in for boost::regex_match) on a batch, in the oracle's output layout:
    -> (caps int32[n][2*ngroups], status uint8[n]) or None when no libpcre can be opened."""
    if isinstance(pattern, str):
        pattern = pattern.encode("utf-8")
    data = np.ascontiguousarray(data, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint32)
    length = np.ascontiguousarray(length, dtype=np.uint32)
    n = len(off)
    caps = np.empty((n, 2 * ngroups), dtype=np.int32)
    status = np.empty((n,), dtype=np.uint8)
    r = lib().orx_pcre_fullmatch_batch(pattern, len(pattern), data.ctypes.data, off.ctypes.data, length.ctypes.data, n, ngroups,
                                       1 if jit else 0, caps.ctypes.data, status.ctypes.data)
    if r == -1:
        return None
    if r < 0:
        raise ValueError("PCRE1 rejects the pattern")
    return caps, status


class OracleRegex:
    """boost::regex stand-in (Perl syntax, dot matches newline, ^/$ multi-line)."""

    def __init__(self, pattern, flags=0):
        if isinstance(pattern, str):
            pattern = pattern.encode("utf-8")
        err = ctypes.create_string_buffer(256)
        self._h = lib().orx_compile(pattern, len(pattern), flags, err, 256)
        if not self._h:
            raise ValueError(err.value.decode())
        self.pattern = pattern
        self.groups = lib().orx_mark_count(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orx_free(self._h)
            self._h = None

    def group_name(self, g):
        r = lib().orx_group_name(self._h, g)
        return r.decode() if r else None

    def fullmatch(self, s):
        """-> list of (begin,end) for groups 0..n, or None (no match); raises on complexity blow-up."""
        caps = (ctypes.c_int32 * (2 * (self.groups + 1)))()
        r = lib().orx_fullmatch(self._h, s, len(s), caps)
        if r < 0:
            raise RuntimeError("complexity exceeded")
        if r == 0:
            return None
        return [(caps[2 * g], caps[2 * g + 1]) for g in range(self.groups + 1)]

    def search(self, s, start=0):
        caps = (ctypes.c_int32 * (2 * (self.groups + 1)))()
        r = lib().orx_search(self._h, s, len(s), start, caps)
        if r < 0:
            raise RuntimeError("complexity exceeded")
        if r == 0:
            return None
        return [(caps[2 * g], caps[2 * g + 1]) for g in range(self.groups + 1)]

    def prefixmatch(self, s):
        """boost::regex_search(..., match_continuous): the match must start at offset 0 (multiline start/continue/end)"""
        caps = (ctypes.c_int32 * (2 * (self.groups + 1)))()
        r = lib().orx_prefixmatch(self._h, s, len(s), caps)
        if r < 0:
            raise RuntimeError("complexity exceeded")
        if r == 0:
            return None
        return [(caps[2 * g], caps[2 * g + 1]) for g in range(self.groups + 1)]

    def fullmatch_batch(self, data, off, length, ngroups=None):
        """data: uint8 ndarray; off/length: uint32 ndarrays.  -> (caps int32[n,2G], status uint8[n])"""
        G = self.groups if ngroups is None else ngroups
        data = np.ascontiguousarray(data, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        length = np.ascontiguousarray(length, dtype=np.uint32)
        n = off.shape[0]
        caps = np.empty((n, 2 * G), dtype=np.int32)
        status = np.empty((n,), dtype=np.uint8)
        lib().orx_fullmatch_batch(self._h, data.ctypes.data, off.ctypes.data, length.ctypes.data, n, G,
                                  caps.ctypes.data, status.ctypes.data)
        return caps, status


    def process_batch(self, data, off, length, keys):
        """The reference processor's per-event work (match + one SetContentNoCopy per key + source tombstone + counters,
        oracle/processor_oracle.c) over a batch; -> dict of the four plugin counters.  Used for the timed CPU baseline."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        length = np.ascontiguousarray(length, dtype=np.uint32)
        blob = b"".join(k.encode() + b"\0" for k in keys)
        counters = (ctypes.c_ulong * 4)()
        lib().orx_process_batch(self._h, data.ctypes.data, off.ctypes.data, length.ctypes.data, off.shape[0], blob,
                                len(keys), counters)
        return dict(zip(("discarded", "out_failed", "out_key_not_found", "out_successful"), [int(c) for c in counters]))
