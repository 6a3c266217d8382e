"""Synthetic log lines for BASELINE.json configs[2] (Grok, 50 patterns): small templates written against the Match patterns
of tests/golden/grok_config3.json (the formats the example_config/processor_grok_patterns files describe), values drawn from
a seeded generator, and EVERY line padded -- in the format's free-text field where it has one, behind the message where it has
none (Grok searches: text around a match does not unmatch it) -- so that line lengths spread log-uniformly over 128..4096 bytes
(mean ~1.1 KB).

Which Match entry a line ends up under is NOT assumed here: tests and bench.py ask the oracle (ordered first-match-wins gives a
line to the earliest pattern that yields something: SYSLOGLINE takes every line that begins with a syslog or ISO-8601 timestamp
and a host, so HAPROXYHTTP / HAPROXYTCP / NETSCREENSESSIONLOG / SHOREWALL / SFW2 can never win in the reference's file order, and
COMBINEDAPACHELOG never wins behind COMMONAPACHELOG).  tests/test_grok_host.py checks that at least 35 of the 50 entries do win."""
import math
import random

SEED = 20260922
PAD = "\x00"     # where a template's padding goes (exactly once per template)

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
    port = lambda r: r.randrange(1, 65536)                       # noqa: E731
    ip4 = lambda r: "%d.%d.%d.%d" % (r.randrange(1, 224), r.randrange(256), r.randrange(256), r.randrange(1, 255))  # noqa: E731
    iface = lambda r: r.choice(["inside", "outside", "dmz", "mgmt"])                                                # noqa: E731
    who = lambda r: r.choice(["Primary", "Secondary"])                                                              # noqa: E731
    hhmmss = lambda r: "%02d:%02d:%02d" % (r.randrange(24), r.randrange(60), r.randrange(60))                       # noqa: E731
    # 0 HTTPD_ERRORLOG (2.0 and 2.4 forms)
    add(lambda r: "[%s %s %02d %s 2014] [%s] [client %s] %s" % (r.choice(["Mon", "Tue", "Wed", "Thu", "Fri"]), r.choice(_MONTHS),
        r.randrange(1, 29), hhmmss(r), r.choice(["error", "warn", "notice"]), ip4(r), PAD))
    add(lambda r: "[%s %s %02d %s.%06d 2014] [core:%s] [pid %d:tid %d] [client %s:%d] AH%05d: %s" % (
        r.choice(["Mon", "Tue", "Wed"]), r.choice(_MONTHS), r.randrange(1, 29), hhmmss(r), r.randrange(10 ** 6),
        r.choice(["error", "info"]), r.randrange(1, 65000), r.randrange(1, 10 ** 9), ip4(r), port(r), r.randrange(99999), PAD))
    # 1 COMMONAPACHELOG (as the head of a combined line: the agent string carries the padding)
    add(lambda r: '%s - %s [%02d/%s/2014:%s +0000] "GET /%s HTTP/1.1" %d %d "http://example.com/%s" "Mozilla/5.0 %s"' % (
        _ip(r), r.choice(["-", "frank", "alice"]), r.randrange(1, 29), r.choice(_MONTHS), hhmmss(r), r.choice(_WORDS[:12]),
        r.choice([200, 301, 404, 500]), r.randrange(10 ** 6), r.choice(_WORDS[:12]), PAD))
    # 3 SYSLOGPAMSESSION, 4 CRONLOG, 5 SYSLOGLINE, 6 SYSLOG5424LINE
    add(lambda r: "%s host%d sshd[%d]: pam_unix(sshd:session): session %s for user %s by %s" % (
        _ts(r), r.randrange(99), r.randrange(1, 65000), r.choice(["opened", "closed"]), r.choice(["root", "deploy", "www-data"]), PAD))
    add(lambda r: "%s host%d CRON[%d]: (root) CMD (%s)" % (_ts(r), r.randrange(99), r.randrange(1, 65000), PAD))
    add(lambda r: "%s host%d %s[%d]: %s" % (_ts(r), r.randrange(99), r.choice(["kernel", "systemd", "dhclient", "postfix/smtpd"]),
                                            r.randrange(1, 65000), PAD))
    add(lambda r: "<%d>1 - host%d.example.com app%d %d ID%d - %s" % (r.randrange(8, 191), r.randrange(99), r.randrange(9),
                                                                     r.randrange(1, 65000), r.randrange(99), PAD))
    # 9 NAGIOSLOGLINE (three of its alternatives)
    add(lambda r: "[%d] SERVICE ALERT: host%d;%s;%s;%s;%d;%s" % (1400000000 + r.randrange(10 ** 7), r.randrange(99),
        r.choice(["HTTP", "PING", "Disk"]), r.choice(["CRITICAL", "WARNING", "OK"]), r.choice(["SOFT", "HARD"]), r.randrange(1, 5), PAD))
    add(lambda r: "[%d] CURRENT SERVICE STATE: host%d;%s;%s;HARD;%d;%s" % (1400000000 + r.randrange(10 ** 7), r.randrange(99),
        r.choice(["HTTP", "PING", "Disk"]), r.choice(["CRITICAL", "WARNING", "OK"]), r.randrange(1, 5), PAD))
    add(lambda r: "[%d] Warning: %s" % (1400000000 + r.randrange(10 ** 7), PAD))
    # 11 CISCO_TAGGED_SYSLOG (no host between the time stamp and the tag: with one the line is SYSLOGLINE's)
    add(lambda r: "<%d>%s: %%ASA-%d-%d: %s" % (r.randrange(100, 191), _ts(r), r.randrange(1, 8), r.randrange(100000, 800000), PAD))
    # 12-18 CISCOFW104001 .. 105009
    add(lambda r: "(%s) Switching to ACTIVE - %s" % (who(r), PAD))
    add(lambda r: "(%s) Switching to STANDBY - %s" % (who(r), PAD))
    add(lambda r: "(%s) Monitoring on interface %s waiting" % (who(r), PAD))
    add(lambda r: "(%s) Monitoring on Interface %s normal" % (who(r), PAD))
    add(lambda r: "(%s) Lost Failover communications with mate on interface %s" % (who(r), PAD))
    add(lambda r: "(%s) Testing Interface %s" % (who(r), PAD))
    add(lambda r: "(%s) Testing on interface %s %s" % (who(r), PAD, r.choice(["Passed", "Failed"])))
    # 19-26
    add(lambda r: "%s TCP connection %s from %s/%d to %s/%d flags SYN ACK on interface %s" % (
        r.choice(["Inbound", "Outbound"]), r.choice(["denied", "permitted"]), _ip(r), port(r), _ip(r), port(r), PAD))
    add(lambda r: "Deny inbound UDP from %s/%d to %s/%d on interface %s -- %s" % (_ip(r), port(r), _ip(r), port(r), iface(r), PAD))
    add(lambda r: "Deny inbound icmp src %s:%s dst %s:%s (type %d, code %d) -- %s" % (iface(r), _ip(r), iface(r), _ip(r),
                                                                                      r.randrange(19), r.randrange(16), PAD))
    add(lambda r: "Deny TCP (no connection) from %s/%d to %s/%d flags RST  on interface %s" % (_ip(r), port(r), _ip(r), port(r), PAD))
    add(lambda r: "Deny UDP reverse path check from %s to %s on interface %s" % (_ip(r), _ip(r), PAD))
    add(lambda r: 'Deny tcp src outside:%s/%d dst inside:%s/%d by access-group "acl_%d" [0x%x, 0x0] -- %s' % (
        _ip(r), port(r), _ip(r), port(r), r.randrange(99), r.randrange(1 << 30), PAD))
    add(lambda r: "access-list acl_%d permitted tcp for user 'u%d' %s/%s(%d) -> %s/%s(%d) hit-cnt %d first hit [0x%x, 0x%x] -- %s" % (
        r.randrange(99), r.randrange(99), iface(r), _ip(r), port(r), iface(r), _ip(r), port(r), r.randrange(1, 999),
        r.randrange(1 << 30), r.randrange(1 << 30), PAD))
    add(lambda r: "access-list acl_%d denied udp %s/%s(%d) -> %s/%s(%d) hit-cnt %d %d-second interval [0x%x, 0x%x] -- %s" % (
        r.randrange(99), iface(r), _ip(r), port(r), iface(r), _ip(r), port(r), r.randrange(1, 999), 300, r.randrange(1 << 30),
        r.randrange(1 << 30), PAD))
    # 27-35
    add(lambda r: "%s Accessed URL %s:/%s" % (_ip(r), _ip(r), PAD))
    add(lambda r: "Failed to locate egress interface for TCP from %s:%s/%d to %s/%d -- %s" % (iface(r), _ip(r), port(r), _ip(r), port(r), PAD))
    add(lambda r: "%d in use, %d most used -- %s" % (r.randrange(100000), r.randrange(100000), PAD))
    add(lambda r: "%s %s TCP connection %d for %s:%s/%d (%s/%d) to %s:%s/%d (%s/%d) -- %s" % (r.choice(["Built", "Teardown"]),
        r.choice(["inbound", "outbound"]), r.randrange(10 ** 7), iface(r), _ip(r), port(r), _ip(r), port(r), iface(r), _ip(r), port(r),
        _ip(r), port(r), PAD))
    add(lambda r: "Built inbound ICMP connection for faddr %s/%d gaddr %s/%d laddr %s/%d -- %s" % (
        _ip(r), r.randrange(9), _ip(r), r.randrange(9), _ip(r), r.randrange(9), PAD))
    add(lambda r: "Built dynamic TCP translation from inside:%s/%d to outside:%s/%d -- %s" % (_ip(r), port(r), _ip(r), port(r), PAD))
    add(lambda r: "Denied ICMP type=%d, code=%d from %s on interface %s -- %s" % (r.randrange(19), r.randrange(16), _ip(r), iface(r), PAD))
    add(lambda r: "No matching connection for ICMP error message: icmp src %s:%s dst %s:%s (type %d, code %d) on %s interface.  "
                  "Original IP payload: udp src %s/%d dst %s/%d -- %s" % (iface(r), _ip(r), iface(r), _ip(r), r.randrange(19),
                                                                         r.randrange(16), iface(r), _ip(r), port(r), _ip(r), port(r), PAD))
    add(lambda r: "Resource 'conns' limit of %d reached for system -- %s" % (r.randrange(1, 10 ** 6), PAD))
    # 36-44
    add(lambda r: "IPSEC: Received a non-IPSec packet (protocol= ICMP) from %s to %s -- %s" % (_ip(r), _ip(r), PAD))
    add(lambda r: "IPSEC: Received an ESP packet (SPI= 0x%x, sequence number= 0x%x) from %s (user= u%d) to %s that failed anti-replay "
                  "checking -- %s" % (r.randrange(1 << 30), r.randrange(1 << 20), _ip(r), r.randrange(99), _ip(r), PAD))
    add(lambda r: "Dropping TCP packet from %s:%s/%d to %s:%s/%d, reason: %s" % (iface(r), _ip(r), port(r), iface(r), _ip(r), port(r), PAD))
    add(lambda r: "Duplicate TCP SYN from %s:%s/%d to %s:%s/%d with different initial sequence number -- %s" % (
        iface(r), _ip(r), port(r), iface(r), _ip(r), port(r), PAD))
    add(lambda r: "Invalid transport field for protocol=UDP, from %s/%d to %s/%d -- %s" % (_ip(r), port(r), _ip(r), port(r), PAD))
    add(lambda r: "IPSEC: An outbound %s SA (SPI= 0x%x) between %s and %s (user= u%d) has been created" % (
        PAD, r.randrange(1 << 30), _ip(r), _ip(r), r.randrange(99)))
    add(lambda r: "TCP access permitted from %s/%d to %s:%s/%d -- %s" % (_ip(r), port(r), iface(r), _ip(r), port(r), PAD))
    add(lambda r: "Group = %s, IP = %s, Automatic NAT Detection Status:     Remote end is NOT behind a NAT device     This   end is "
                  "behind a NAT device" % (PAD, _ip(r)))
    add(lambda r: "[ Scanning] drop rate-%d exceeded. Current burst rate is %d per second, max configured rate is %d; "
                  "Current average rate is %d per second, max configured rate is %d; Cumulative total count is %d -- %s" % (
                      r.randrange(1, 3), r.randrange(999), r.randrange(999), r.randrange(999), r.randrange(999), r.randrange(10 ** 6), PAD))
    # 45 SHOREWALL's format (in the reference's order SYSLOGLINE wins it: a line with the structure, searched by both)
    add(lambda r: "%s fw%d kernel: [%d.%d] Shorewall:net2fw:DROP:IN=eth0 OUT= MAC=00:11:22:33:44:55:66:77:88:99:aa:bb:08:00 "
                  "SRC=%s DST=%s LEN=%d TOS=0x00 PREC=0x00 TTL=%d ID=%d DF PROTO=TCP SPT=%d DPT=%d WINDOW=1024 RES=0x00 SYN "
                  "URGP=0 %s" % (_ts(r), r.randrange(9), r.randrange(10 ** 6), r.randrange(10 ** 6), ip4(r), ip4(r),
                                 r.randrange(40, 1500), r.randrange(1, 255), r.randrange(65536), port(r), port(r), PAD))
    # 47 JAVASTACKTRACEPART, 48 CATALINALOG, 49 TOMCATLOG
    add(lambda r: "    at com.example.%s.%s.handle(Handler%d.java:%d) -- %s" % (r.choice(_WORDS[:12]), r.choice(_WORDS[:12]),
                                                                                r.randrange(99), r.randrange(1, 4000), PAD))
    add(lambda r: "%s %d, 2014 %d:%02d:%02d %s org.apache.catalina.%s.Runner%d %s" % (
        r.choice(_MONTHS), r.randrange(1, 29), r.randrange(1, 13), r.randrange(60), r.randrange(60), r.choice(["AM", "PM"]),
        r.choice(_WORDS[:12]), r.randrange(9), PAD))
    add(lambda r: "2014-%02d-%02d %02d:%02d:%02d,%03d -0700 | %s | org.apache.tomcat.%s.Pool - %s" % (
        r.randrange(1, 13), r.randrange(1, 29), r.randrange(24), r.randrange(60), r.randrange(60), r.randrange(1000),
        r.choice(["ERROR", "WARN", "INFO", "DEBUG"]), r.choice(_WORDS[:12]), PAD))
    return T


MIN_LINE, MAX_LINE = 128, 4096
_T = None


def grok_lines(n, seed=SEED, unmatched=0.05):
    """-> list of n byte strings of MIN_LINE..MAX_LINE bytes, lengths log-uniform.  About `unmatched` of them are free text that no
    log format should take."""
    global _T
    if _T is None:
        _T = _templates()
    r = random.Random(seed)
    out = []
    for _ in range(n):
        target = int(math.exp(r.uniform(math.log(MIN_LINE), math.log(MAX_LINE))))
        if r.random() < unmatched:
            line = ("~~ " + _tail(r, target) + " ~~").encode("utf-8")
        else:
            fmt = r.choice(_T)(r)
            base = len(fmt.encode("utf-8")) - 1
            line = fmt.replace(PAD, _tail(r, max(8, target - base + 1))).encode("utf-8")
        # (the generators overshoot by a word: cut to the target, never below the format's own text; a cut inside a 2-byte
        # character leaves a stray byte, which the byte-oriented engines take like any other)
        out.append(line[:max(MIN_LINE, min(MAX_LINE, max(target, 0)))] if len(line) > MAX_LINE else line)
    return out
