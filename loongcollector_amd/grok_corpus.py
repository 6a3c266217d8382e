"""Synthetic log lines for BASELINE.json configs[2] (Grok, 50 patterns): small templates, one family per group of Match
patterns of tests/golden/grok_config3.json, values drawn from a seeded generator, and -- where the format ends in free
text -- a tail padded so that line lengths spread log-uniformly over 128..4096 bytes.

Which Match entry a line ends up under is NOT assumed here: tests and bench.py ask the oracle (ordered first-match-wins
can give a line to an earlier, more general pattern)."""
import math
import random

SEED = 20260922

_WORDS = ["connection", "timeout", "user", "session", "cache", "retry", "backend", "frontend", "worker", "request", "queue",
          "handshake", "teardown", "interface", "policy", "denied", "accepted", "checksum", "fragment", "overflow", "λ", "ok"]
_MONTHS = ["Jan", "Feb", "Mar", "Apr", "May", "Jun", "Jul", "Aug", "Sep", "Oct", "Nov", "Dec"]


def _ip(r):
    if r.random() < 0.15:
        return "2001:db8:%x:%x::%x" % (r.randrange(65536), r.randrange(65536), r.randrange(65536))
    return "%d.%d.%d.%d" % (r.randrange(1, 224), r.randrange(256), r.randrange(256), r.randrange(1, 255))


def _ts(r):
    return "%s %2d %02d:%02d:%02d" % (r.choice(_MONTHS), r.randrange(1, 29), r.randrange(24), r.randrange(60), r.randrange(60))


def _tail(r, target):
    out = []
    n = 0
    while n < target:
        w = r.choice(_WORDS) + (str(r.randrange(1000)) if r.random() < 0.3 else "")
        out.append(w)
        n += len(w.encode("utf-8")) + 1
    return " ".join(out)


def _templates():
    T = []
    add = T.append
    add(lambda r, t: "%s host%d CRON[%d]: (root) CMD (%s)" % (_ts(r), r.randrange(99), r.randrange(1, 65000), _tail(r, t)))
    add(lambda r, t: "(%s) Switching to ACTIVE - %s" % (r.choice(["Primary", "Secondary"]), _tail(r, t)))
    add(lambda r, t: "(%s) Monitoring on interface %s waiting" % (r.choice(["Primary", "Secondary"]), _tail(r, t)))
    add(lambda r, t: "%s TCP connection %s from %s/%d to %s/%d flags SYN ACK on interface %s" % (
        r.choice(["Inbound", "Outbound"]), r.choice(["denied", "permitted"]), _ip(r), r.randrange(65536), _ip(r),
        r.randrange(65536), _tail(r, t)))
    add(lambda r, t: "Deny TCP (no connection) from %s/%d to %s/%d flags RST  on interface %s" % (
        _ip(r), r.randrange(65536), _ip(r), r.randrange(65536), _tail(r, t)))
    add(lambda r, t: "Deny UDP reverse path check from %s to %s on interface %s" % (_ip(r), _ip(r), _tail(r, t)))
    add(lambda r, t: 'Deny tcp src outside:%s/%d dst inside:%s/%d by access-group "acl_%d" [0x%x, 0x0]' % (
        _ip(r), r.randrange(65536), _ip(r), r.randrange(65536), r.randrange(99), r.randrange(1 << 30)))
    add(lambda r, t: "%d in use, %d most used" % (r.randrange(100000), r.randrange(100000)))
    add(lambda r, t: "Built inbound ICMP connection for faddr %s/%d gaddr %s/%d laddr %s/%d" % (
        _ip(r), r.randrange(9), _ip(r), r.randrange(9), _ip(r), r.randrange(9)))
    add(lambda r, t: "Built dynamic TCP translation from inside:%s/%d to outside:%s/%d" % (
        _ip(r), r.randrange(65536), _ip(r), r.randrange(65536)))
    add(lambda r, t: "IPSEC: Received a non-IPSec packet (protocol= ICMP) from %s to %s" % (_ip(r), _ip(r)))
    add(lambda r, t: "Invalid transport field for protocol=UDP, from %s/%d to %s/%d" % (
        _ip(r), r.randrange(65536), _ip(r), r.randrange(65536)))
    add(lambda r, t: "[ Scanning] drop rate-%d exceeded. Current burst rate is %d per second, max configured rate is %d; "
                     "Current average rate is %d per second, max configured rate is %d; Cumulative total count is %d" % (
                         r.randrange(1, 3), r.randrange(999), r.randrange(999), r.randrange(999), r.randrange(999),
                         r.randrange(10 ** 6)))
    add(lambda r, t: "    at com.example.%s.%s.handle(Handler%d.java:%d)" % (r.choice(_WORDS[:12]), r.choice(_WORDS[:12]),
                                                                             r.randrange(99), r.randrange(1, 4000)))
    add(lambda r, t: "%s %d, 2014 %d:%02d:%02d %s org.apache.catalina.%s.Runner%d %s" % (
        r.choice(_MONTHS), r.randrange(1, 29), r.randrange(1, 13), r.randrange(60), r.randrange(60), r.choice(["AM", "PM"]),
        r.choice(_WORDS[:12]), r.randrange(9), _tail(r, t)))
    add(lambda r, t: "2014-%02d-%02d %02d:%02d:%02d,%03d -0700 | %s | org.apache.tomcat.%s.Pool - %s" % (
        r.randrange(1, 13), r.randrange(1, 29), r.randrange(24), r.randrange(60), r.randrange(60), r.randrange(1000),
        r.choice(["ERROR", "WARN", "INFO", "DEBUG"]), r.choice(_WORDS[:12]), _tail(r, t)))
    add(lambda r, t: "%s fw%d kernel: [%d.%d] Shorewall:net2fw:DROP:IN=eth0 OUT= MAC=00:11:22:33:44:55:66:77:88:99:aa:bb:08:00 "
                     "SRC=%s DST=%s LEN=%d TOS=0x00 PREC=0x00 TTL=%d ID=%d DF PROTO=TCP SPT=%d DPT=%d WINDOW=1024 RES=0x00 SYN "
                     "URGP=0 %s" % (_ts(r), r.randrange(9), r.randrange(10 ** 6), r.randrange(10 ** 6),
                                    "%d.%d.%d.%d" % (r.randrange(1, 224), r.randrange(256), r.randrange(256), r.randrange(1, 255)),
                                    "%d.%d.%d.%d" % (r.randrange(1, 224), r.randrange(256), r.randrange(256), r.randrange(1, 255)),
                                    r.randrange(40, 1500), r.randrange(1, 255), r.randrange(65536), r.randrange(65536),
                                    r.randrange(65536), _tail(r, t)))
    add(lambda r, t: '%s - %s [%02d/%s/2014:%02d:%02d:%02d +0000] "GET /%s HTTP/1.1" %d %d' % (
        _ip(r), r.choice(["-", "frank", "alice"]), r.randrange(1, 29), r.choice(_MONTHS), r.randrange(24), r.randrange(60),
        r.randrange(60), _tail(r, min(t, 900)).replace(" ", "/"), r.choice([200, 301, 404, 500]), r.randrange(10 ** 6)))
    return T


def grok_lines(n, seed=SEED, unmatched=0.05):
    """-> list of n byte strings.  About `unmatched` of them are free text that no log format should take."""
    r = random.Random(seed)
    T = _templates()
    out = []
    for _ in range(n):
        target = int(math.exp(r.uniform(math.log(128), math.log(4096))))
        if r.random() < unmatched:
            out.append(("~~ " + _tail(r, target) + " ~~").encode("utf-8")[:4096])
            continue
        line = r.choice(T)(r, max(0, target - 100)).encode("utf-8")
        out.append(line[:4096])
    return out
