"""Test-only replay of the NFA program for patterns with atomic groups / possessive quantifiers: the same step the
kernel's atomic path takes (nfa_kernel.hpp: nfaAtomicStep), written out in Python so that the packed tables and the
commit rules can be checked without a GPU.  Not part of the product.

Threads carry their unsettled atomic-segment memberships ("lineage": [g, seg, exited]); a step
  1. walks ALL epsilon paths of all threads in priority order, viable or not (tdfa.cpp commitAtomic has the rules:
     leaving a group closes its segment for everything of lower priority except continuations of the very same exit),
  2. drops memberships nobody can act on any more,
  3. drops a thread that mirrors a higher-priority one on the same position (same rule as the TDFA builder's
     pairwise `mirrors`; with empty lineages this is the ordinary "first thread on a position wins").
"""
import numpy as np

from loongcollector_amd import binding as B
from tests.helpers.table_interp import NfaInterp, _with_run_captures

ASSERT_EVENT = 20000
MAX_LINEAGE = 6      # nfa_kernel.hpp kNfaLineage
MAX_THREADS = 64


class AtomicNfaInterp(NfaInterp):
    def __init__(self, rx):
        super().__init__(rx)
        blob = rx.table(B.LC_TABLE_NFA_BLOB, np.uint32)
        self.atomic = int(blob[14]) != 0
        self.events = []          # per position: per path: [(code, visit), ...]
        if not self.atomic:
            return
        npaths = int(blob[9])
        pe = blob[int(blob[7]) // 4:int(blob[7]) // 4 + 2 * npaths].reshape(-1, 2)[:, 1]   # path word y: events
        ev = blob[int(blob[15]) // 4:]
        fs = blob[int(blob[6]) // 4:int(blob[6]) // 4 + self.npos + 2]
        for p in range(self.npos + 1):
            lst = []
            for i in range(int(fs[p]), int(fs[p + 1])):
                start, cnt = int(pe[i]) >> 8, int(pe[i]) & 0xFF
                one = []
                for w in ev[start:start + cnt]:
                    code = int(w) & 0xFFFF
                    one.append((code - 65536 if code >= 32768 else code, int(w) >> 16))
                lst.append(one)
            self.events.append(lst)

    # ---- one priority-ordered commit pass; cands: (target, src, tags, lin, events, target_ok)
    @staticmethod
    def _commit(cands, holds, step):
        kept, closed = [], {}

        for target, src, tags, lin0, ev, target_ok in cands:
            def exit_visit_for(g, frm):
                depth = 0
                for code, visit in ev[frm:]:
                    if code >= ASSERT_EVENT:
                        if not (holds >> (code - ASSERT_EVENT)) & 1:
                            return 0
                    elif code == g + 1:
                        depth += 1
                    elif code == -(g + 1):
                        if depth == 0:
                            return visit
                        depth -= 1
                return 0

            dead = False
            for g, seg, ex in lin0:
                cl = closed.get((g, seg))
                if cl is not None and (ex or cl[0] != src or exit_visit_for(g, 0) != cl[1]):
                    dead = True
            if dead:
                continue
            lin = [list(e) for e in lin0]
            ok = True
            for i, (code, visit) in enumerate(ev):
                if code >= ASSERT_EVENT:
                    if not (holds >> (code - ASSERT_EVENT)) & 1:
                        ok = False
                        break
                elif code > 0:
                    g = code - 1
                    seg = (step + 1) * 64 + src          # one fresh segment per (step, source thread, group)
                    cl = closed.get((g, seg))
                    if cl is not None and (cl[0] != src or exit_visit_for(g, i + 1) != cl[1]):
                        dead = True
                        break
                    lin.append([g, seg, False])
                else:
                    g = -code - 1
                    for e in reversed(lin):
                        if e[0] == g and not e[2]:
                            e[2] = True
                            closed.setdefault((g, e[1]), (src, visit))
                            break
            if dead or not ok or not target_ok:
                continue
            kept.append([target, src, tags, lin])
        # memberships nobody can act on any more
        for i, k in enumerate(kept):
            k[3] = [e for e in k[3] if not e[2] or any(
                f[0] == e[0] and f[1] == e[1] and not f[2] for j in range(i) for f in kept[j][3])]

        def holds_(c, e, inside_only):
            return any(x[0] == e[0] and x[1] == e[1] and (not inside_only or not x[2]) for x in c[3])

        def mirrors(hi, lo):
            if kept[hi][0] != kept[lo][0]:
                return False
            for e in kept[hi][3]:
                if not holds_(kept[lo], e, False) and any(holds_(kept[k], e, True) for k in range(hi)):
                    return False
            for e in kept[lo][3]:
                if not e[2] and not holds_(kept[hi], e, True) and any(holds_(kept[k], e, False)
                                                                       for k in range(lo + 1, len(kept))):
                    return False
            return True

        changed = True
        while changed:
            changed = False
            for i in range(1, len(kept)):
                if any(mirrors(j, i) for j in range(i)):
                    del kept[i]
                    changed = True
                    break
        return kept

    @_with_run_captures
    def fullmatch(self, s, max_threads=MAX_THREADS, start=0):
        if not self.atomic:
            return super().fullmatch(s, max_threads=max_threads, start=start)
        threads = [(self.npos, [-1] * self.nslots, [])]      # (position, caps, lineage)
        prev_cls = self.ncls
        if start:
            threads = [(0, [-1] * self.nslots, [])]
            prev_cls = int(self.cmap[s[start - 1]])
        for pos in range(start, len(s)):
            cls = int(self.cmap[s[pos]])
            holds = self.behind[prev_cls] | self.ahead[cls]
            prev_cls = cls
            cands = []
            for t, (p, _, lin) in enumerate(threads):
                for k, (tgt, cond, tags) in enumerate(self.follow[p]):
                    ok = tgt >= 0 and (self.posmask[tgt] >> cls) & 1
                    cands.append((tgt, t, tags, lin, self.events[p][k], bool(ok)))
            kept = self._commit(cands, holds, pos)
            if len(kept) > max_threads or any(len(k[3]) > MAX_LINEAGE for k in kept):
                return "overflow"
            new = []
            for tgt, src, tags, lin in kept:
                caps = list(threads[src][1])
                for sl in range(self.nslots):
                    if (tags >> sl) & 1:
                        caps[sl] = pos
                new.append((tgt, caps, lin))
            threads = new
            if not threads:
                return None
        holds = self.behind[prev_cls] | self.ahead[self.ncls]
        cands = []
        for t, (p, _, lin) in enumerate(threads):
            for k, (tgt, cond, tags) in enumerate(self.follow[p]):
                cands.append((-1 - len(cands), t, tags, lin, self.events[p][k], tgt < 0))   # unique "positions"
        kept = self._commit(cands, holds, len(s))
        if not kept:
            return None
        _, src, tags, _ = kept[0]
        caps = list(threads[src][1])
        for sl in range(self.nslots):
            if (tags >> sl) & 1:
                caps[sl] = len(s)
        return caps
