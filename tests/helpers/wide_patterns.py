"""Values for the four example_config Grok patterns that need the extensions of the NFA engine: %{HTTPD_ERRORLOG} (68 byte
classes: 4-word class masks), %{HAPROXYHTTP} (53 groups = 106 capture slots: 4 tag words per aux entry, NS=128 kernel),
%{NAGIOSLOGLINE} (139 groups = 278 slots: 10 tag words, NS=320 kernel) and %{SYSLOGPAMSESSION} (a run capture,
"(?=%{GREEDYDATA:message})").  Well-formed lines of each format plus seeded one-byte mutations of them; below, patterns and
subjects for run captures on their own."""
import random

WIDE_PATTERNS = ["HTTPD_ERRORLOG", "HAPROXYHTTP"]

_LINES = [
    b"[Mon Aug 23 15:25:35 2010] [error] [client 80.154.42.54] File does not exist: /var/www/phpmy-admin",
    b"[Sat Apr 25 06:00:13.246302 2015] [core:error] [pid 1234:tid 140333] [client 10.1.1.1:5000] AH00126: Invalid URI in "
    b"request GET /x",
    b"[Sat Apr 25 06:00:13.246302 2015] [proxy:warn] [pid 99] (111)Connection refused: AH00957: HTTP: attempt to connect to "
    b"127.0.0.1:8080 (localhost) failed",
    b"[Mon Aug 23 15:25:35 2010] [notice] Apache/2.2.15 (Unix) configured -- resuming normal operations",
    b"Feb  6 12:14:14 localhost haproxy[14389]: 10.0.1.2:33317 [06/Feb/2009:12:14:14.655] http-in static/srv1 10/0/30/69/109 "
    b"200 2750 - - ---- 1/1/1/1/0 0/0 {1wt.eu|10.9.8.7|en|http://r.example/|curl/7.1} {text/html|gzip|no-cache|Mon} "
    b"\"GET /index.html HTTP/1.1\"",
    b"Dec  9 13:01:26 lb1 haproxy[28029]: 127.0.0.1:39759 [09/Dec/2013:12:59:46.633] loadbalancer default/instance8 "
    b"0/51536/1/48082/99627 200 83285 - - ---- 87/87/87/1/0 0/67 \"GET /path/to/image HTTP/1.1\"",
    b"2013-12-09T13:01:26+01:00 lb1 haproxy[28029]: 127.0.0.1:39759 [09/Dec/2013:12:59:46.633] lb default/i8 "
    b"0/5/1/4/9 503 0 - - SC-- 8/8/8/1/0 0/6 \"<BADREQ>\"",
    b"nothing to see here",
]


def wide_values(mutations=6, seed=3):
    r = random.Random(seed)
    out = list(_LINES)
    for line in _LINES:
        for _ in range(mutations):
            b = bytearray(line)
            k = r.randrange(len(b))
            op = r.random()
            if op < 0.4:
                b[k] = r.choice(b" x[]:9\"/|")
            elif op < 0.7:
                del b[k]
            else:
                b.insert(k, r.choice(b" x[]:9|"))
            out.append(bytes(b))
    return out


# "(?=(S*))" run captures (regex_ast.hpp Node::runCapture): patterns and subjects shared by the CPU and GPU tests
RUN_CAPTURE_PATTERNS = [rb"(\w+) (?=(.*))(\w+)\((\d+)\)", rb"a(?=([0-9]*))\d*(x?)", rb"(?=.*)abc", rb"(?:(?=([a-z]*))[a-z]{2},)+",
                        rb"(?=(?:([^\n]*)))a.*", rb"(?=(?P<rest>[^;]*))(?:(\w+)=(\d+);?)+"]
RUN_CAPTURE_SUBJECTS = [b"pam su(12)", b"pam su(12", b"a123x", b"a12", b"abc", b"ab,cd,", b"ab,c,", b"pam su\nx(1)", b"a\nb", b"",
                        b"k=1;kk=22;x", b"k=1;kk=22", b"a" + b"7" * 300 + b"x"]

_PAM = [
    b"Jul 11 13:30:01 ip-10-0-0-1 CRON[26176]: pam_unix(cron:session): session opened for user root by (uid=0)",
    b"Jul 11 13:30:01 host sshd[1]: pam_unix(sshd:session): session closed for user alice",
    b"Jul 11 13:30:01 host sshd[1]: pam_unix(sshd:session): session closed for user alice\nsecond line",
    b"<13>Jul  1 03:00:00 host su: pam_unix(su:session): session opened for user bob by alice(uid=1000)",
    b"Jul 11 13:30:01 host su: pam_unix(su:session) session opened for user",
]
WIDE_PATTERNS.append("SYSLOGPAMSESSION")
_LINES.extend(_PAM)

_NAGIOS = [
    b"[1427925600] CURRENT HOST STATE: nagios.example.com;UP;HARD;1;PING OK - Packet loss = 0%, RTA = 2.24 ms",
    b"[1427925600] CURRENT SERVICE STATE: nagios.example.com;HTTP;OK;HARD;1;HTTP OK: HTTP/1.1 200 OK - 453 bytes",
    b"[1427925689] SERVICE ALERT: varnish.example.com;Varnish Backend Connections;CRITICAL;SOFT;1;Current value: 154.0",
    b"[1427956600] SERVICE NOTIFICATION: nagiosadmin;ntp.example.com;NTP;OK;notify-service-by-email;NTP OK: Offset 0.000339 secs",
    b"[1427955600] HOST NOTIFICATION: nagiosadmin;db1.example.com;DOWN;notify-host-by-email;CRITICAL - Host Unreachable",
    b"[1427955600] TIMEPERIOD TRANSITION: 24x7;-1;1",
    b"[1427925600] LOG ROTATION: DAILY",
    b"[1427925600] Warning: Return code of 127 for check of service 'x' on host 'y' was out of bounds.",
    b"[1427925600] EXTERNAL COMMAND: SCHEDULE_SVC_DOWNTIME;host1;svc;1427925600;1427929200;1;0;3600;admin;patching",
    b"[1427925600] PASSIVE SERVICE CHECK: host1;disk;0;DISK OK - free space: / 3326 MB (56%)",
    b"[142792560x] SERVICE ALERT: a;b;CRITICAL;SOFT;1;c",
]
WIDE_PATTERNS.append("NAGIOSLOGLINE")
_LINES.extend(_NAGIOS)
