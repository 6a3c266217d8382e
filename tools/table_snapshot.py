#!/usr/bin/env python3
"""Hash every device table of a fixed pattern set (CPU only): the 50 Grok entries of configs[2] searched and anchored, regex A / B in
four modes, the 702 golden patterns full-match and search -- 1 508 handles, every LC_TABLE_* of each.
    python tools/table_snapshot.py out.json [path/to/liblc_regex_gpu.so]
    python tools/table_snapshot.py --diff a.json b.json
Two snapshots, one taken with the library before a change to the host compilers (regex parser, follow NFA, tagged-DFA construction,
packers) and one after, show whether the change left the tables bit-identical -- how the round-4 rewrite of buildTdfa's inner loop
was accepted."""
import ctypes, hashlib, json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) == 4 and sys.argv[1] == "--diff":
    a, b = json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
    diff = [k for k in sorted(set(a) | set(b)) if a.get(k, {}).get("tables") != b.get(k, {}).get("tables") or a.get(k, {}).get("rc") != b.get(k, {}).get("rc")]
    print(len(a), "handles;", len(diff), "differ")
    for k in diff:
        print(" ", k, "|", a.get(k, {}).get("rc"), a.get(k, {}).get("err", "")[:60], "->", b.get(k, {}).get("rc"), b.get(k, {}).get("err", "")[:60])
    print("compile seconds: %.1f -> %.1f" % (sum(v["s"] for v in a.values()), sum(v["s"] for v in b.values())))
    sys.exit(1 if diff else 0)
if len(sys.argv) > 2: os.environ["LC_REGEX_GPU_LIB"] = sys.argv[2]
from loongcollector_amd import binding as B, corpus
L = B.load()
L.lc_regex_compile.restype = ctypes.c_int
L.lc_regex_compile.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p, ctypes.c_size_t]
L.lc_regex_table.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
L.lc_regex_free.argtypes = [ctypes.c_void_p]
SEARCH, NAMED, NODOT, NOML, RE2, PREFIX = 1<<5, 1<<4, 1<<1, 1<<2, 1<<6, 1<<7
GROK = SEARCH|NAMED|NODOT|NOML|RE2
jobs = []
from loongcollector_amd.grok import Grok
_cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "grok_config3.json"), encoding="utf-8"))
_g = Grok(Match=_cfg["match"], CustomPatterns=_cfg["custom_patterns"], AnchoredFirst=False)
pats = [_g.expanded(i) for i in range(len(_cfg["match"]))]
for i, p in enumerate(pats):
    jobs.append(("grok%02d.search" % i, p, GROK, 0))
    jobs.append(("grok%02d.anchored" % i, p, GROK | PREFIX, 1))
jobs.append(("A", corpus.REGEX_A, 0, 0)); jobs.append(("B", corpus.REGEX_B, 0, 0))
jobs.append(("A.search", corpus.REGEX_A, SEARCH, 0)); jobs.append(("B.prefix", corpus.REGEX_B, PREFIX, 0))
g = json.load(open(os.path.join(ROOT, "tests", "golden", "regex_golden.json")))
seen = set()
for c in (g["cases"] if isinstance(g, dict) else g):
    p = c.get("p") if isinstance(c, dict) else None
    if p and p not in seen:
        seen.add(p); jobs.append(("golden%03d:" % len(seen) + p[:40], p, 0, 0)); jobs.append(("goldenS%03d:" % len(seen) + p[:40], p, SEARCH, 0))
def run(job):
    name, pat, flags, eng = job
    p = pat.encode("utf-8") if isinstance(pat, str) else pat
    h = ctypes.c_void_p(); err = ctypes.create_string_buffer(512)
    t0 = time.time()
    rc = L.lc_regex_compile(p, len(p), flags, eng, ctypes.byref(h), err, 512)
    out = {"rc": rc, "err": err.value.decode()[:200], "s": round(time.time() - t0, 2), "tables": {}}
    if h:
        for which in range(12):
            d = ctypes.c_void_p(); n = ctypes.c_size_t()
            if L.lc_regex_table(h, which, ctypes.byref(d), ctypes.byref(n)) == 0 and n.value:
                out["tables"][str(which)] = [n.value, hashlib.sha256(ctypes.string_at(d, n.value)).hexdigest()[:16]]
        L.lc_regex_free(h)
    return name, out
t0 = time.time()
with ThreadPoolExecutor(8) as ex:
    res = dict(ex.map(run, jobs))
json.dump(res, open(sys.argv[1], "w"), indent=0, sort_keys=True)
print(len(res), "patterns,", round(time.time() - t0, 1), "s; compile seconds total", round(sum(v["s"] for v in res.values()), 1))
