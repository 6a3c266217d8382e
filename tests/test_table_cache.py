"""Compiled automata across process restarts (csrc/table_cache.cpp; include/lc_regex_gpu.h lc_runtime_set_table_cache_dir; Grok config key
"CacheDir").  Each case runs in an interpreter of its own: the cache directory is process-wide state."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNAP = r"""
import ctypes, hashlib, json, sys, time
from loongcollector_amd import binding as B
from loongcollector_amd.grok import Grok
cfg = json.load(open("tests/golden/grok_config3.json"))
names = ["%{CISCOFW106001}", "%{CRONLOG}", "%{SYSLOG5424LINE}", "%{CISCOFW313005}", "%{CATALINALOG}", "%{CISCOFW402117}"]
t0 = time.time()
g = Grok(Match=names, CustomPatterns=cfg["custom_patterns"], **({"CacheDir": sys.argv[1]} if len(sys.argv) > 1 and sys.argv[1] else {}))
g.wait_ready()
dt = time.time() - t0
L = B.load()
L.lc_regex_table.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
F = B.LC_SYNTAX_SEARCH | B.LC_SYNTAX_NAMED_ONLY | B.LC_SYNTAX_NO_DOTALL | B.LC_SYNTAX_NO_MULTILINE | B.LC_SYNTAX_REGEXP2
tables = {}
for i in range(len(names)):
    for flags in (F, F | B.LC_SYNTAX_PREFIX):
        try:
            rx = B.GpuRegex(g.expanded(i).encode(), syntax_flags=flags)
        except B.RegexUnsupportedError as e:
            tables["%d/%x" % (i, flags)] = "refused: " + str(e)[:80]
            continue
        row = [rx.info()["engine"], rx.info()["states"]]
        for which in range(12):
            d = ctypes.c_void_p(); n = ctypes.c_size_t()
            if L.lc_regex_table(rx.handle, which, ctypes.byref(d), ctypes.byref(n)) == 0 and n.value:
                row.append(hashlib.sha256(ctypes.string_at(d, n.value)).hexdigest()[:16])
        tables["%d/%x" % (i, flags)] = row
st = (ctypes.c_uint64 * 4)(); L.lc_runtime_table_cache_stats(st)
print(json.dumps({"seconds": dt, "tables": tables, "stats": list(st)}))
"""


def _run(code, *args, env=None):
    e = dict(os.environ)
    e.pop("LC_TABLE_CACHE_DIR", None)
    e.update(env or {})
    out = subprocess.run([sys.executable, "-c", code, *args], cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_second_load_comes_from_the_cache_with_identical_tables(tmp_path):
    """six entries of configs[2] (two that determinise only anchored, one that does not determinise at all, one LDS-size): without a
    cache, with an empty cache, with the cache filled -- the same tables bit for bit, and the third load does no construction"""
    cache = str(tmp_path / "tables")
    plain = _run(SNAP, "")
    assert plain["stats"] == [0, 0, 0, 0]                       # off by default: nothing looked up, nothing written
    assert not os.path.exists(cache)
    first = _run(SNAP, cache)
    assert first["tables"] == plain["tables"]
    assert first["stats"][1] > 20 and first["stats"][2] > 20, first["stats"]   # looked up, built, stored
    files = sorted(os.listdir(cache))
    assert len(files) >= 20 and all(f.startswith("lc_tdfa_") and f.endswith(".bin") for f in files), files[:5]
    second = _run(SNAP, cache)
    assert second["tables"] == plain["tables"]
    hits, misses, stored, failures = second["stats"]
    assert misses == 0 and stored == 0 and hits > 10 and failures >= 1, second["stats"]   # (the verdicts of failed constructions too)
    assert second["seconds"] < 1.5 and second["seconds"] < first["seconds"] / 4, (first["seconds"], second["seconds"])
    # the environment variable does what the config key does
    third = _run(SNAP, "", env={"LC_TABLE_CACHE_DIR": cache})
    assert third["tables"] == plain["tables"] and third["stats"][1] == 0 and third["stats"][0] > 10


def test_damaged_or_foreign_cache_files_are_ignored(tmp_path):
    cache = str(tmp_path / "tables")
    first = _run(SNAP, cache)
    files = sorted(os.listdir(cache))
    # truncate one, fill one with noise, make one empty: every one of them is rebuilt, none becomes tables
    with open(os.path.join(cache, files[0]), "r+b") as f:
        f.truncate(max(1, os.path.getsize(os.path.join(cache, files[0])) // 2))
    with open(os.path.join(cache, files[1]), "wb") as f:
        f.write(os.urandom(4096))
    with open(os.path.join(cache, files[2]), "wb") as f:
        pass
    # round 6 (ADVICE): ONE flipped bit deep inside a table (the file keeps its length and its header: only the checksum can tell), a
    # truncated file padded back to its length, and a whole, valid file under ANOTHER key's name (the key travels in the file)
    big = sorted(files[3:], key=lambda f: -os.path.getsize(os.path.join(cache, f)))
    with open(os.path.join(cache, big[0]), "r+b") as f:
        size = os.path.getsize(os.path.join(cache, big[0]))
        f.seek(size // 2)
        b = f.read(1)
        f.seek(size // 2)
        f.write(bytes([b[0] ^ 0x10]))
    with open(os.path.join(cache, big[1]), "r+b") as f:
        size = os.path.getsize(os.path.join(cache, big[1]))
        f.truncate(size - 64)
        f.seek(0, 2)
        f.write(b"\0" * 64)
    with open(os.path.join(cache, big[2]), "rb") as f:
        foreign = f.read()
    with open(os.path.join(cache, big[3]), "wb") as f:
        f.write(foreign)
    again = _run(SNAP, cache)
    assert again["tables"] == first["tables"]
    assert again["stats"][1] >= 6 and again["stats"][2] >= 6, again["stats"]   # looked up, found unusable, rebuilt and stored again
    final = _run(SNAP, cache)
    assert final["tables"] == first["tables"] and final["stats"][1] == 0


def test_the_stamp_follows_the_sources_that_shape_the_tables_and_the_ab_switches_bypass_the_cache(tmp_path):
    """the cache key's build stamp was table_cache.cpp's own compile time -- an edit of tdfa.cpp alone did not change it.
    build.py now hashes every table-shaping source into obj/table_sources_stamp.inc; and a process under LC_TDFA_NO_DSE /
    LC_TDFA_NO_MINIMIZE (they change the construction's output) neither reads nor writes the cache."""
    import hashlib
    from loongcollector_amd import build as native_build
    native_build.build_native()
    inc = os.path.join(native_build.LIBDIR, "obj", "table_sources_stamp.inc")
    assert os.path.exists(inc)
    hh = hashlib.sha256()
    for name in native_build.TABLE_SHAPING_SOURCES:
        with open(os.path.join(native_build.CSRC, name), "rb") as f:
            hh.update(name.encode() + b"\0" + f.read() + b"\0")
    assert open(inc).read().strip() == '"lc-tables-%s"' % hh.hexdigest()[:32]
    assert {"tdfa.cpp", "screen_dfa.cpp", "tdfa.hpp", "table_cache.cpp"} <= set(native_build.TABLE_SHAPING_SOURCES)
    out = subprocess.run([sys.executable, "-c", "import ctypes\nfrom loongcollector_amd import binding as B\nL = B.load()\n"
                          "L.lc_runtime_table_cache_stamp.restype = ctypes.c_char_p\nprint(L.lc_runtime_table_cache_stamp().decode())"],
                         cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.stdout.strip() == open(inc).read().strip().strip('"'), out.stderr[-500:]   # the library that is loaded carries this stamp
    cache = str(tmp_path / "tables")
    switched = _run(SNAP, cache, env={"LC_TDFA_NO_MINIMIZE": "1"})
    assert switched["stats"] == [0, 0, 0, 0] and not os.path.exists(cache)
