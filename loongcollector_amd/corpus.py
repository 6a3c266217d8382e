"""Deterministic synthetic log corpora for BASELINE.json's configs (bench + parity tests).

Field recipe follows the reference's own benchmark generator
(test/engine/trigger/log/remote_file_benchmark.py:27: `ip - - [time] "METHOD url HTTP/1.1" status bytes "referer" "ua"`)
and SURVEY.md section 8(d): every line is padded (URL and user-agent, alphabet [A-Za-z0-9/%._-]) to exactly
`line_bytes` bytes excluding the trailing '\n', and every line fully matches its regex (regex_match semantics).
Lines are laid out back to back with '\n' separators, i.e. exactly what one LogFileReader read buffer holds
after ProcessorSplitLogStringNative has sliced it (core/plugin/processor/inner/ProcessorSplitLogStringNative.cpp:130-160).
"""
import numpy as np

SEED = 20260921

# 10 groups: docs/cn/plugins/processor/native/processor-parse-regex-native.md:58
REGEX_A = (r'([\d\.]+) \S+ \S+ \[(\S+) \S+\] \"(\w+) ([^\\"]*)\" ([\d\.]+) (\d+) (\d+) (\d+|-) '
           r'\"([^\\"]*)\" \"([^\\"]*)\"')
KEYS_A = ["ip", "time", "method", "url", "request_time", "request_length", "status", "length", "ref_url", "browser"]
# 11 groups: test/benchmark/local/test_cases/performance_file_to_blackhole_loongcollector/loongcollector.yaml:9
REGEX_B = r'^([^ ]*) ([^ ]*) ([^ ]*) \[([^\]]*)\] "(\S+) ([^\"]*) (\S*)" ([^ ]*) ([^ ]*) "([^\"]*)" "([^\"]*)"'
KEYS_B = ["ip", "ident", "auth", "timestamp", "method", "request", "http_version", "response_code", "bytes",
          "referrer", "user_agent"]

_PAD = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789/%._-", dtype=np.uint8)
_METHODS = [b"GET"] * 14 + [b"POST"] * 4 + [b"PUT", b"DELETE"]
_STATUS = [b"200"] * 16 + [b"304", b"404", b"500", b"502"]
_MONTHS = [b"Jan", b"Feb", b"Mar", b"Apr", b"May", b"Jun", b"Jul", b"Aug", b"Sep", b"Oct", b"Nov", b"Dec"]


def _pad(rng, n):
    return _PAD[rng.integers(0, len(_PAD), size=n)].tobytes()


def _one_line(rng, kind, line_bytes):
    ip = b"%d.%d.%d.%d" % tuple(int(x) for x in rng.integers(1, 255, size=4))
    ts = b"%02d/%s/%04d:%02d:%02d:%02d" % (int(rng.integers(1, 29)), _MONTHS[int(rng.integers(0, 12))],
                                             int(rng.integers(2020, 2027)), int(rng.integers(0, 24)),
                                             int(rng.integers(0, 60)), int(rng.integers(0, 60)))
    method = _METHODS[int(rng.integers(0, len(_METHODS)))]
    status = _STATUS[int(rng.integers(0, len(_STATUS)))]
    nbytes = b"%d" % int(rng.integers(1, 10001))
    if kind == "A":
        head = ip + b" - - [" + ts + b" +0800] \"" + method + b" /"
        mid = b"\" %d.%03d %d " % (int(rng.integers(0, 10)), int(rng.integers(0, 1000)), int(rng.integers(1, 100000)))
        mid += status + b" " + (nbytes if rng.integers(0, 10) else b"-") + b" \"https://example.com/"
        tail = b"\" \"Mozilla/5.0 "
        end = b"\""
    else:
        head = ip + b" - - [" + ts + b" +0000] \"" + method + b" /"
        mid = b" HTTP/1.1\" " + status + b" " + nbytes + b" \"https://example.com/"
        tail = b"\" \"Mozilla/5.0 "
        end = b"\""
    fixed = len(head) + len(mid) + len(tail) + len(end)
    free = line_bytes - fixed
    if free < 3:
        raise ValueError("line_bytes too small")
    ref = int(rng.integers(0, min(40, free - 2)))
    rest = free - ref
    url = int(rng.integers(1, rest))
    ua = rest - url
    line = head + _pad(rng, url) + mid + _pad(rng, ref) + tail + _pad(rng, ua) + end
    assert len(line) == line_bytes, (len(line), line_bytes)
    return line


def apache_pool(kind="A", pool_lines=8192, line_bytes=512, seed=SEED):
    """-> uint8 array [pool_lines, line_bytes] of distinct fully-matching lines."""
    rng = np.random.Generator(np.random.MT19937(seed + (0 if kind == "A" else 1)))
    pool = np.empty((pool_lines, line_bytes), dtype=np.uint8)
    for i in range(pool_lines):
        pool[i] = np.frombuffer(_one_line(rng, kind, line_bytes), dtype=np.uint8)
    return pool


def blank_referrer(line: bytes) -> bytes:
    """The same combined-format line with an EMPTY referrer field ("" -- its capture group matches the empty string) and the
    user agent lengthened so that the line keeps its length."""
    q = [j for j, c in enumerate(line) if c == 0x22]              # the quotes: ... "R" "U"
    r0, r1, u0, u1 = q[-4], q[-3], q[-2], q[-1]
    tail = b'"" "' + b"x" * ((r1 - r0 - 1) + (u1 - u0 - 1)) + b'"'
    assert len(tail) == u1 - r0 + 1
    return line[:r0] + tail + line[u1 + 1:]


def apache_batch(n_lines, kind="A", line_bytes=512, seed=SEED, pool_lines=8192, poison_every=0, empty_every=0):
    """Build one batch: (data uint8[n*(line_bytes+1)], off uint32[n+1], len uint32[n]).

    Lines are drawn (with replacement, seeded) from a pool of `pool_lines` distinct lines and joined with '\n'.
    off has n+1 entries so that len[i] == off[i+1]-off[i]-1 (the separator).  poison_every=k replaces every k-th
    line's first byte by '{' so that it must FAIL to match (failure-path coverage); empty_every=k gives every k-th line
    an empty referrer field (blank_referrer).
    """
    pool = apache_pool(kind, pool_lines, line_bytes, seed)
    rng = np.random.Generator(np.random.MT19937(seed + 7919))
    idx = rng.integers(0, pool_lines, size=n_lines)
    buf = np.empty((n_lines, line_bytes + 1), dtype=np.uint8)
    buf[:, :line_bytes] = pool[idx]
    buf[:, line_bytes] = 10
    if empty_every:
        for i in range(0, n_lines, empty_every):
            buf[i, :line_bytes] = np.frombuffer(blank_referrer(buf[i, :line_bytes].tobytes()), dtype=np.uint8)
    if poison_every:
        buf[::poison_every, 0] = ord("{")
    off = (np.arange(n_lines + 1, dtype=np.uint64) * (line_bytes + 1)).astype(np.uint32)
    length = np.full((n_lines,), line_bytes, dtype=np.uint32)
    return buf.reshape(-1), off, length


def apache_lines(n_lines, kind="A", line_bytes=512, seed=SEED, poison_every=0):
    """The headline corpus by SURVEY.md section 8(d)'s recipe: EVERY line generated on its own from one std::mt19937_64 stream (seed
    20260921), field distributions as stated there (tools/corpus_gen.cpp; apache_batch draws from a pool of 8 192 numpy-made lines
    instead, which the round-4 review flagged).  Same layout as apache_batch: (data uint8[n * (line_bytes + 1)], off uint32[n + 1],
    len uint32[n]); poison_every=k makes every k-th line fail to match."""
    import ctypes
    from . import build as _build
    lib = ctypes.CDLL(_build.build_corpus_gen())
    lib.lc_corpus_apache_lines.restype = ctypes.c_int
    lib.lc_corpus_apache_lines.argtypes = [ctypes.c_char, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
    buf = np.empty((n_lines, line_bytes + 1), dtype=np.uint8)
    rc = lib.lc_corpus_apache_lines(kind.encode(), n_lines, line_bytes, seed + (0 if kind == "A" else 1), buf.ctypes.data)
    if rc != 0:
        raise ValueError("line_bytes too small")
    if poison_every:
        buf[::poison_every, 0] = ord("{")
    off = (np.arange(n_lines + 1, dtype=np.uint64) * (line_bytes + 1)).astype(np.uint32)
    length = np.full((n_lines,), line_bytes, dtype=np.uint32)
    return buf.reshape(-1), off, length


def mixed_batch(n_lines, seed=SEED + 3, min_len=128, max_len=2048, json_fraction=0.3):
    """Config-5 style corpus: nginx-combined lines (match REGEX_B) mixed with JSON lines (must fail), log-uniform
    lengths.  -> (data, off[n+1], len[n])"""
    rng = np.random.Generator(np.random.MT19937(seed))
    lens = np.exp(rng.uniform(np.log(min_len), np.log(max_len), size=n_lines)).astype(np.int64)
    chunks = []
    for i in range(n_lines):
        L = int(lens[i])
        if rng.random() < json_fraction:
            body = b'{"time":"2024-06-25T23:59:59Z","level":"info","msg":"'
            line = body + _pad(rng, max(1, L - len(body) - 2)) + b'"}'
        else:
            line = _one_line(rng, "B", max(L, 160))
        chunks.append(line)
    length = np.array([len(c) for c in chunks], dtype=np.uint32)
    off = np.zeros(n_lines + 1, dtype=np.uint32)
    off[1:] = np.cumsum(length.astype(np.uint64) + 1).astype(np.uint32)
    data = np.frombuffer(b"\n".join(chunks) + b"\n", dtype=np.uint8).copy()
    return data, off, length


# ---- multi-line logs (Java stack traces): what the multiline processors are configured for in practice
MULTILINE_START = r"\d{4}-\d{2}-\d{2} \d{2}:\d{2}:\d{2}.*"
_ML_CLASSES = ["com.example.order.OrderService", "org.springframework.web.servlet.DispatcherServlet", "java.util.concurrent.ThreadPoolExecutor",
               "io.netty.channel.AbstractChannelHandlerContext", "com.zaxxer.hikari.pool.HikariPool", "org.apache.catalina.core.StandardWrapperValve"]


def multiline_buffer(target_bytes, seed=SEED + 17, unmatched_head=0):
    """-> bytes: logs of one start line ("2024-.. ERROR ...") followed by 0..40 stack-frame lines ("\tat ..."), sometimes a
    "Caused by:" section; `unmatched_head` frame lines come first (a read buffer that begins inside a log).  Ends with '\n'."""
    import random
    rng = random.Random(seed)
    out = []
    size = 0
    for _ in range(unmatched_head):
        out.append(b"\tat %s.run(%s.java:%d)" % (rng.choice(_ML_CLASSES).encode(), b"Worker", rng.randint(1, 999)))
        size += len(out[-1]) + 1
    sec = 0
    while size < target_bytes:
        sec += rng.randint(0, 3)
        head = b"2024-01-%02d %02d:%02d:%02d.%03d %s [%s-%d] %s - %s" % (
            1 + sec // 86400 % 28, sec // 3600 % 24, sec // 60 % 60, sec % 60, rng.randint(0, 999),
            rng.choice([b"ERROR", b"WARN", b"INFO"]), rng.choice([b"http-nio-8080-exec", b"pool-1-thread", b"main"]), rng.randint(1, 64),
            rng.choice(_ML_CLASSES).encode(), b"request %d failed: timeout after %d ms" % (rng.randint(1, 10 ** 6), rng.randint(10, 30000)))
        out.append(head)
        size += len(head) + 1
        frames = int(rng.choice([0, 0, 1, 3, 8, 15, 25, 40]) * rng.random())
        for k in range(frames):
            if k and rng.random() < 0.05:
                ln = b"Caused by: java.lang.IllegalStateException: " + rng.choice(_ML_CLASSES).encode()
            else:
                ln = b"\tat %s.%s(%s.java:%d)" % (rng.choice(_ML_CLASSES).encode(), rng.choice([b"invoke", b"run", b"doFilter", b"process"]),
                                                   rng.choice([b"Service", b"Filter", b"Handler"]), rng.randint(1, 2000))
            out.append(ln)
            size += len(ln) + 1
    return b"\n".join(out) + b"\n"
