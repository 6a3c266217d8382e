"""oracle/filter_oracle.py -- CPU restatement of processor_filter_regex_native.  TEST INFRASTRUCTURE ONLY (oracle/README.md).

Follows core/plugin/processor/ProcessorFilterNative.cpp of the reference: Init precedence :30-157 (ConditionExp >
FilterKey+FilterRegex > Include), ProcessEvent :178-216, IsMatched :258-286, expression nodes :381-486, and the
non-UTF-8 blanking routine noneUtf8 :297-379.  Regex leaves are boost::regex_match restated by oracle/bt_regex.c.
Events are dicts {"contents": {key: value bytes}, ...}; contents keep insertion order like LogEvent's content list.
Pinned on the reference's own unit-test vectors (ProcessorFilterNativeUnittest.cpp:184-560, transcribed in
tests/golden/filter_vectors.json)."""
from oracle.oracle import OracleRegex


def none_utf8(b: bytes):
    """-> (is_bad, blanked copy) as ProcessorFilterNative::noneUtf8 defines it"""
    s = bytearray(b)
    n, i, bad_any = len(s), 0, False
    cont = lambda k: (s[k] & 0xC0) == 0x80
    while i < n:
        c = s[i]
        need, bad = 1, False
        if c & 0x80 == 0:
            pass
        elif c & 0xE0 == 0xC0:
            need = 2
            if i + 1 >= n or not cont(i + 1):
                bad = True
            else:
                u = ((c & 0x1F) << 6) | (s[i + 1] & 0x3F)
                bad = not (0x80 <= u <= 0x7FF)
        elif c & 0xF0 == 0xE0:
            need = 3
            if i + 2 >= n or not cont(i + 1) or not cont(i + 2):
                bad = True
            else:
                u = (((c & 0x0F) << 12) | ((s[i + 1] & 0x3F) << 6) | (s[i + 2] & 0x3F)) & 0xFFFF
                bad = not (u >= 0x800)
        elif c & 0xF8 == 0xF0:
            need = 4
            if i + 3 >= n or not cont(i + 1) or not cont(i + 2) or not cont(i + 3):
                bad = True
            else:
                u = ((c & 0x07) << 18) | ((s[i + 1] & 0x3F) << 12) | ((s[i + 2] & 0x3F) << 6) | (s[i + 3] & 0x3F)
                bad = not (0x10000 <= u <= 0x10FFFF)
        else:
            bad = True
        if bad:
            bad_any = True
            s[i] = 0x20
            i += 1
            continue
        i += need
    return bad_any, bytes(s)


class FilterOracle:
    def __init__(self, config):
        self.mode = "bypass"
        self.root = None
        self.rule = []
        ce = config.get("ConditionExp")
        if ce is not None:
            if not isinstance(ce, dict):
                raise ValueError("object param ConditionExp is not of type object")
            self.root = self._parse(ce)
            if self.root is None:
                raise ValueError("object param ConditionExp is not valid")
            self.mode = "expression"
        if self.mode == "bypass":
            keys, regs = config.get("FilterKey", []), config.get("FilterRegex", [])
            if len(keys) != len(regs):
                raise ValueError("param FilterKey and FilterRegex does not have the same size")
            if keys:
                self.rule = [(k, OracleRegex(r)) for k, r in zip(keys, regs)]
                self.mode = "rule"
        if self.mode == "bypass" and config.get("Include"):
            self.rule = [(k, OracleRegex(r)) for k, r in config["Include"].items()]
            self.mode = "rule"
        self.discard_non_utf8 = bool(config.get("DiscardingNonUTF8", False))

    def _parse(self, v):
        if not isinstance(v, dict):
            return None
        if isinstance(v.get("operator"), str) and isinstance(v.get("operands"), list):
            op, ops = v["operator"].lower(), v["operands"]
            if op not in ("not", "and", "or"):
                return None
            if op == "not" and len(ops) == 1:
                c = self._parse(ops[0])
                return None if c is None else ("not", c)
            if op in ("and", "or") and len(ops) == 2:
                l, r = self._parse(ops[0]), self._parse(ops[1])
                return None if l is None or r is None else (op, l, r)
            return None
        if (isinstance(v.get("key"), str) and isinstance(v.get("exp"), str)) or not isinstance(v.get("type"), str):
            t = v.get("type") if isinstance(v.get("type"), str) else ""
            if t.lower() != "regex":
                return None
            key = v.get("key") if isinstance(v.get("key"), str) else ""
            exp = v.get("exp") if isinstance(v.get("exp"), str) else ""
            return ("leaf", key, OracleRegex(exp))
        return None

    def _eval(self, node, contents):
        if node[0] == "leaf":
            return node[1] in contents and node[2].fullmatch(contents[node[1]]) is not None
        if node[0] == "not":
            return not self._eval(node[1], contents)
        if node[0] == "and":
            return self._eval(node[1], contents) and self._eval(node[2], contents)
        return self._eval(node[1], contents) or self._eval(node[2], contents)

    def process_event(self, contents):
        """contents: dict key(str) -> value(bytes), insertion-ordered.  -> (keep, contents)"""
        res = True
        if self.mode == "expression":
            res = bool(contents) and self._eval(self.root, contents)
        elif self.mode == "rule":
            res = bool(contents) and all(k in contents and r.fullmatch(contents[k]) is not None for k, r in self.rule)
        if res and self.discard_non_utf8:
            out, renamed = dict(contents), []
            for k, v in contents.items():
                bad, fixed = none_utf8(v)
                if bad:
                    out[k] = v = fixed
                kb = k.encode("utf-8", "surrogateescape")
                bad, fixed = none_utf8(kb)
                if bad:
                    renamed.append((fixed.decode("utf-8", "surrogateescape"), v))
                    del out[k]
            for k, v in renamed:
                out[k] = v
            contents = out
        return res, contents

    def process(self, events):
        out = []
        for contents in events:
            keep, c = self.process_event(contents)
            if keep:
                out.append(c)
        return out
