"""Test-only model of the DECIDE kernel (csrc/nfa_decide_kernel.hpp): a depth-first, priority-ordered walk of the packed
follow-NFA program -- what a backtracking matcher (boost, regexp2) does, on the tables the device engines run.  The
breadth-first kernels give up on a line when their thread list overflows (LC_OVERFLOW); this walk decides such lines
with no bound on live threads.  Written out in Python so that the rules can be checked against the golden vectors and
against the breadth-first interpreters without a GPU.  Not part of the product.

Every transition of the follow NFA consumes one byte, so depth d of the walk is input offset start+d: a frame per
depth holds (position, next path to try).  Atomic groups are a CUT: a path that leaves group instance S drops every
choice made inside S -- frames opened after S's entry frame are exhausted, the entry frame loses its alternatives that
enter the same group, and the leaving frame keeps only continuations of that very exit (same exit visit); exactly
nfaAtomicStep's rules, but in their natural depth-first form.  A memo nibble per (position, offset) makes the walk
linear: 1 = "fails from here", 1+j = "fails after committing the j innermost enclosing groups" (the cut is replayed).
"""
import numpy as np

from tests.helpers.nfa_atomic_interp import ASSERT_EVENT, AtomicNfaInterp
from tests.helpers.table_interp import _with_run_captures

OLD, BY_SELF, BY_OTHER = 0, 1, 2


class DfsNfaInterp(AtomicNfaInterp):
    def __init__(self, rx):
        super().__init__(rx)
        self.steps = 0

    @_with_run_captures
    def fullmatch(self, s, start=0, memo=True, budget=None):
        L = len(s)
        ncls, npos = self.ncls, self.npos
        cls = [int(self.cmap[b]) for b in s]

        def holds_at(i):
            prev = ncls if i == 0 else cls[i - 1]
            nxt = ncls if i == L else cls[i]
            return self.behind[prev] | self.ahead[nxt]

        def exit_visit_for(ev, frm, g, holds):
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

        nframes = L - start + 1
        pos = [0] * nframes
        q = [0] * nframes            # next path to try; q[d] - 1 = the path taken while frame d + 1 is alive
        exhausted = [False] * nframes  # a cut dropped the frame's remaining alternatives
        chain_top = [-1] * nframes
        chain_depth = [0] * nframes
        arena_mark = [0] * nframes
        cut_level = [0] * nframes
        closed = [[] for _ in range(nframes)]
        nodes = []          # [g, depth, parent, cdepth, cut_done]
        memo_tab = {} if memo else None

        def cut(n, d, real, v):
            g, d0, _, cdepth_n, done = nodes[n]
            if d0 < d:
                if not done:
                    for dd in range(d0 + 1, d):
                        exhausted[dd] = True
                        cut_level[dd] = max(cut_level[dd], chain_depth[dd] - cdepth_n + 1)
                    closed[d0].append((g, BY_OTHER, 0))
                    nodes[n][4] = True
                cut_level[d] = max(cut_level[d], chain_depth[d] - cdepth_n + 1)
                if real:
                    if (g, OLD, v) not in closed[d]:
                        closed[d].append((g, OLD, v))
                else:
                    exhausted[d] = True
            else:
                closed[d].append((g, BY_SELF, v) if real else (g, BY_OTHER, 0))

        pos[0] = npos if start == 0 else 0
        d = 0
        self.steps = 0
        while d >= 0:
            i = start + d
            p = pos[d]
            lst = self.follow[p]
            if exhausted[d] or q[d] >= len(lst):
                if memo_tab is not None and cut_level[d] <= 14:
                    memo_tab[(p, i)] = 1 + cut_level[d]
                d -= 1
                continue
            k = q[d]
            q[d] += 1
            self.steps += 1
            if budget is not None and self.steps > budget:
                return "gave_up"
            del nodes[arena_mark[d]:]
            cur, cur_depth = chain_top[d], chain_depth[d]
            tgt, cond, tags = lst[k]
            holds = holds_at(i)
            dead = False
            ok = True
            if self.atomic:
                ev = self.events[p][k]
                for g, kind, v in closed[d]:
                    if kind == OLD and exit_visit_for(ev, 0, g, holds) != v:
                        dead = True
                if dead:
                    continue
                for idx, (code, visit) in enumerate(ev):
                    if code >= ASSERT_EVENT:
                        if not (holds >> (code - ASSERT_EVENT)) & 1:
                            ok = False
                            break
                    elif code > 0:
                        g = code - 1
                        for g2, kind, v in closed[d]:
                            if g2 != g:
                                continue
                            if kind == BY_OTHER or (kind == BY_SELF and exit_visit_for(ev, idx + 1, g, holds) != v):
                                dead = True
                        if dead:
                            break
                        nodes.append([g, d, cur, cur_depth + 1, False])
                        cur, cur_depth = len(nodes) - 1, cur_depth + 1
                    else:
                        g = -code - 1
                        n = cur
                        while n != -1 and nodes[n][0] != g:
                            n = nodes[n][2]
                        if n == -1:
                            continue
                        cut(n, d, True, visit)
                        cur, cur_depth = nodes[n][2], nodes[n][3] - 1
            else:
                ok = (cond & ~holds) == 0
            if dead or not ok:
                continue
            if i == L:
                if tgt >= 0:
                    continue
                caps = [-1] * self.nslots
                for dd in range(d):
                    _, _, tg = self.follow[pos[dd]][q[dd] - 1]
                    for sl in range(self.nslots):
                        if (tg >> sl) & 1:
                            caps[sl] = start + dd
                for sl in range(self.nslots):
                    if (tags >> sl) & 1:
                        caps[sl] = L
                return caps
            if tgt < 0 or not (self.posmask[tgt] >> cls[i]) & 1:
                continue
            m = memo_tab.get((tgt, i + 1), 0) if memo_tab is not None else 0
            if m == 0:
                d += 1
                pos[d] = tgt
                q[d] = 0
                exhausted[d] = False
                chain_top[d], chain_depth[d] = cur, cur_depth
                arena_mark[d] = len(nodes)
                cut_level[d] = 0
                closed[d] = []
            elif m > 1:
                n = cur
                for _ in range(m - 2):
                    n = nodes[n][2]
                cut(n, d, False, 0)
        return None
