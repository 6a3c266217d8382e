"""Test-only interpreters for the compiled device tables (TDFA tables / follow NFA blob).

They let the `-m "not gpu"` suite check the host compilers (regex_parse -> follow_nfa -> tdfa) against the
oracle and the golden vectors without a GPU.  They are NOT part of the product and are never imported by
loongcollector_amd/: the product has no CPU execution path.
"""
import numpy as np

from loongcollector_amd import binding as B


def _with_run_captures(fullmatch):
    """groups written "(?=(S*))": the tables stamp the begin slot only; the end is where the run of S bytes that starts
    there ends (gpu_runtime.hip run_capture_kernel does this after the match kernel)"""
    def wrapped(self, s, *args, **kw):
        caps = fullmatch(self, s, *args, **kw)
        if isinstance(caps, list):
            for g, members in self.runs:
                b = caps[2 * g]
                if b >= 0:
                    e = b
                    while e < len(s) and s[e] in members:
                        e += 1
                    caps[2 * g + 1] = e
        return caps
    return wrapped


class TdfaInterp:
    def __init__(self, rx):
        hdr = rx.table(B.LC_TABLE_TDFA_HEADER, np.uint32)
        assert hdr is not None, "pattern has no TDFA"
        self.runs = rx.run_captures()
        self.nstates, self.ncls, self.nregs, self.nslots, self.start = [int(x) for x in hdr[:5]]
        self.cmap = rx.table(B.LC_TABLE_CLASSMAP, np.uint8)
        self.trans = rx.table(B.LC_TABLE_TDFA_TRANS, np.uint32)
        self.ops_start = rx.table(B.LC_TABLE_TDFA_OPSSTART, np.uint32)
        self.ops = rx.table(B.LC_TABLE_TDFA_OPS, np.uint16)
        self.final_id = rx.table(B.LC_TABLE_TDFA_FINALID, np.uint16)
        self.final_map = rx.table(B.LC_TABLE_TDFA_FINALMAP, np.uint8)
        self.start_after = rx.table(B.LC_TABLE_TDFA_STARTAFTER, np.uint32)  # None unless a search pattern

    @_with_run_captures
    def fullmatch(self, s: bytes, start=0):
        """-> flat caps [b1,e1,b2,e2,...] for groups 1..G, or None.  start > 0 (search patterns only): resume the search
        at that offset, seeing the byte before it (what the kernels do for lc_regex_match_device_from)."""
        state = self.start if start == 0 else int(self.start_after[int(self.cmap[s[start - 1]])])
        regs = [-1] * (self.nregs + 1)
        for pos, b in enumerate(s):
            if pos < start:
                continue
            t = int(self.trans[state * self.ncls + int(self.cmap[b])])
            lst = t >> 16
            if lst:
                o = int(self.ops_start[lst])
                n = int(self.ops[o])
                for k in range(n):
                    w = int(self.ops[o + 1 + k])
                    dst, src = w & 0xFF, w >> 8
                    regs[dst] = pos if src == 0xFF else regs[src]
            state = t & 0xFFFF
            if state == 0:
                return None
        fid = int(self.final_id[state])
        if fid == 0xFFFF:
            return None
        out = []
        for sl in range(self.nslots):
            m = int(self.final_map[fid * self.nslots + sl])
            out.append(len(s) if m == 0xFF else (-1 if m == 0xFE else regs[m]))
        return out


class NfaInterp:
    """Ordered-thread-list simulation of the packed NFA blob (what the wave-per-line kernel does)."""

    def __init__(self, rx):
        blob = rx.table(B.LC_TABLE_NFA_BLOB, np.uint32)
        assert blob is not None
        self.runs = rx.run_captures()
        self.npos, self.nslots, self.ncls = int(blob[1]), int(blob[2]), int(blob[3])
        raw = blob.view(np.uint8)
        self.cmap = raw[int(blob[4]):int(blob[4]) + 256].copy()
        mw, aw = int(blob[20]), int(blob[21])                          # NF_MASK_WORDS, NF_AUX_WORDS
        assert mw == (4 if self.ncls > 64 else 2) and aw == (16 if self.nslots > 128 else 8 if self.nslots > 64 else 4)
        pm = blob[int(blob[5]) // 4:int(blob[5]) // 4 + mw * self.npos]
        self.posmask = [sum(int(pm[mw * p + k]) << (32 * k) for k in range(mw)) for p in range(self.npos)]
        fs = blob[int(blob[6]) // 4:int(blob[6]) // 4 + self.npos + 2]
        paths = blob[int(blob[7]) // 4:int(blob[7]) // 4 + 2 * int(blob[9])].reshape(-1, 2)   # NF_OFF_PATHS
        aux = blob[int(blob[17]) // 4:].reshape(-1)                                            # NF_OFF_AUX (aw words each)
        ncl = self.ncls
        self.behind = [int(x) for x in blob[int(blob[12]) // 4:int(blob[12]) // 4 + ncl + 1]]   # NF_OFF_BEHIND
        self.ahead = [int(x) for x in blob[int(blob[13]) // 4:int(blob[13]) // 4 + ncl + 1]]    # NF_OFF_AHEAD
        # NF_OFF_STABLE: the steady-state fast path of the kernel (bit c: on byte class c a thread on this position only ever
        # repeats itself -- or, in a search pattern, also re-spawns the wrapper's suffix thread, which changes nothing)
        st = blob[int(blob[11]) // 4:int(blob[11]) // 4 + mw * (self.npos + 1)]
        self.stable = [sum(int(st[mw * p + k]) << (32 * k) for k in range(mw)) for p in range(self.npos + 1)]
        self.search_suffix = self.npos - 1 if int(blob[22]) else -1                                # NF_SUFFIX (search and anchored search)
        # NF_OFF_QUASI: doomed spawns -- (position, class c, next class d) on which the thread only repeats itself
        self.quasi_idx, self.quasi_rows = None, None
        if int(blob[23]):
            q = int(blob[23])
            nrows, rows_at = int(blob[q // 4]), int(blob[q // 4 + 1])
            self.quasi_idx = raw[q + 8:q + 8 + 2 * (self.npos + 1)].view(np.uint16)
            rows = blob[rows_at // 4:rows_at // 4 + nrows * self.ncls * mw]
            self.quasi_rows = [[sum(int(rows[(r * self.ncls + c) * mw + k]) << (32 * k) for k in range(mw)) for c in range(self.ncls)]
                               for r in range(nrows)]
        self.follow = []
        for p in range(self.npos + 1):
            lst = []
            for i in range(int(fs[p]), int(fs[p + 1])):
                x = int(paths[i][0])
                tgt, a = x & 0xFFFF, x >> 16
                cond = int(aux[aw * a])
                tags = sum(int(aux[aw * a + 1 + k]) << (32 * k) for k in range(min(aw - 1, 10)))
                lst.append((-1 if tgt == 0xFFFF else tgt, cond, tags))
            self.follow.append(lst)

    @_with_run_captures
    def fullmatch(self, s: bytes, max_threads=64, start=0, steady=True):
        """start > 0 (search patterns only): resume at that offset -- one thread on the wrapper's prefix position
        (position 0), having just consumed the byte before the resume point.  steady: take the kernel's steady-state fast path
        (a byte on which every live thread is steady is skipped) and its suffix pruning, as nfa_match_kernel does."""
        threads = [(self.npos, [-1] * self.nslots)]  # (position, caps)
        prev_cls = self.ncls                     # edge entry: start of input
        if start:
            threads = [(0, [-1] * self.nslots)]
            prev_cls = int(self.cmap[s[start - 1]])
        for pos, b in enumerate(s):
            if pos < start:
                continue
            # (the kernel stops when the suffix thread is the only one left: it takes every byte and ends on MATCH)
            if steady and self.search_suffix >= 0 and len(threads) == 1 and threads[0][0] == self.search_suffix:
                break
            cls = int(self.cmap[b])
            nxt = int(self.cmap[s[pos + 1]]) if pos + 1 < len(s) else -1

            def quiet(p):
                if (self.stable[p] >> cls) & 1:
                    return True
                if nxt < 0 or self.quasi_idx is None or p >= len(self.quasi_idx) or not int(self.quasi_idx[p]):
                    return False
                return bool((self.quasi_rows[int(self.quasi_idx[p]) - 1][cls] >> nxt) & 1)
            if steady and all(quiet(p) for p, _ in threads):
                prev_cls = cls
                continue
            holds = self.behind[prev_cls] | self.ahead[cls]
            new, seen = [], set()
            for p, caps in threads:
                for tgt, cond, tags in self.follow[p]:
                    if tgt < 0 or tgt in seen or not (self.posmask[tgt] >> cls) & 1:
                        continue
                    if cond & ~holds:
                        continue
                    seen.add(tgt)
                    c2 = list(caps)
                    for sl in range(self.nslots):
                        if (tags >> sl) & 1:
                            c2[sl] = pos
                    new.append((tgt, c2))
            if steady and self.search_suffix >= 0:      # nothing ranked below a thread on the suffix position can win any more
                for k, (p, _) in enumerate(new):
                    if p == self.search_suffix:
                        new = new[:k + 1]
                        break
            if len(new) > max_threads:
                return "overflow"
            threads = new
            prev_cls = cls
            if not threads:
                return None
        holds = self.behind[prev_cls] | self.ahead[self.ncls]
        for p, caps in threads:
            for tgt, cond, tags in self.follow[p]:
                if tgt >= 0 or (cond & ~holds):
                    continue
                c2 = list(caps)
                for sl in range(self.nslots):
                    if (tags >> sl) & 1:
                        c2[sl] = len(s)
                return c2
        return None


class TdfaBlobInterp:
    """Walks the PACKED TDFA tables exactly as the kernel addresses them (device_tables.h): the low half of a transition
    entry is the LDS address of the next row, the high half the byte offset of the stamped register (or a move-list id);
    `compact` = the tables of the opt-in COMPACT kernel variant the pattern was compiled for (LC_TDFA_COMPACT=256|512|1024:
    16-bit registers; 1024 = byte-indexed rows), LC_TABLE_TDFA_WIDE_BLOB."""

    def __init__(self, rx, compact=False):
        blob = rx.table(B.LC_TABLE_TDFA_WIDE_BLOB if compact else B.LC_TABLE_TDFA_BLOB, np.uint32)
        assert blob is not None
        self.blob, self.raw = blob, blob.view(np.uint8)
        (self.nstates, self.ncls, self.nregs, self.nslots, self.start_row, off_after, _, off_finalid, off_finalmap,
         off_opsstart, off_ops, _, self.row_bytes, self.id_col, self.block) = [int(x) for x in blob[1:16]]
        # TD_NREGS: low half = registers incl. dummy; bits 16..28 = offset/16 of the fold words; bit 31 = no general program
        fold_off = ((self.nregs >> 16) & 0x1FFF) * 16
        self.no_general = bool(self.nregs >> 31)
        self.nregs &= 0xFFFF
        self.fold = [int(w) for w in blob[fold_off // 4 + 1:fold_off // 4 + 1 + int(blob[fold_off // 4])]] if fold_off else None
        self.compact = compact
        self.wide = self.row_bytes == 257 * 4            # rows indexed by the byte itself
        assert not self.wide or (compact and self.block == 1024)
        self.reg_stride = self.block * (2 if compact else 4)
        self.cmap = self.raw[64:320]
        self.final_id = self.raw[off_finalid:off_finalid + 2 * self.nstates].view(np.uint16)
        self.final_map = self.raw[off_finalmap:]
        self.ops_start = blob[off_opsstart // 4:]
        self.ops = self.raw[off_ops:off_ops + (len(self.raw) - off_ops) // 2 * 2].view(np.uint16)
        self.start_after = blob[off_after // 4:off_after // 4 + self.ncls] if off_after else None
        self.runs = rx.run_captures()

    @_with_run_captures
    def fullmatch(self, s: bytes, start=0):
        t = self.start_row if start == 0 else int(self.start_after[int(self.cmap[s[start - 1]]) >> 2])
        regs = {r: 0 for r in range(self.nregs)} if self.fold is not None else {}   # (fold words: registers start at 0)
        limit = 0xFFFF if self.compact else 0xFFFFFFFF
        for pos in range(start, len(s)):
            col = s[pos] * 4 if self.wide else int(self.cmap[s[pos]])
            e = int(self.blob[((t & 0xFFFF) + col) // 4])
            t, f = e & 0xFFFF, e >> 16
            if f & 1:                                       # general move list
                assert not self.no_general
                o = int(self.ops_start[f >> 1])
                for w in self.ops[o + 1:o + 1 + int(self.ops[o])]:
                    dst, src = int(w) & 0xFF, int(w) >> 8
                    regs[dst] = (pos - start) & limit if src == 0xFF else regs.get(src, -1)
            else:
                assert f % self.reg_stride == 0 and f // self.reg_stride < self.nregs
                regs[f // self.reg_stride] = (pos - start) & limit
        for w in self.fold or ():                           # a member reads as max(member, its set's register)
            for k in (1, 2, 3):
                r = (w >> (8 * k)) & 0xFF
                if r != 0xFF:
                    regs[r] = max(regs[r], regs[w & 0xFF])
        state = ((t & 0xFFFF) - 320) // self.row_bytes
        fid = int(self.final_id[state])
        if state == 0 or fid == 0xFFFF:
            return None
        caps = []
        for sl in range(self.nslots):
            m = int(self.final_map[fid * self.nslots + sl])
            caps.append(len(s) if m == 0xFF else -1 if m == 0xFE else regs.get(m, -1) + start)
        return caps


class TdfaPair1Interp(TdfaBlobInterp):
    """The ONE-STAMP byte-pair table of a compact blob (LC_TDFA_PAIR=2; device_tables.h TP1_*), walked as the pair kernel walks it:
    pairs and 8-byte chunks are aligned in MEMORY (`head` = the line's offset in its first aligned 16 bytes), a pair entry stamps
    rA in line (value pos, or pos + 1 with the DELTA bit), the second register of a DOUBLE entry is settled behind its chunk as
    max(register, pos + 1), registers start at 0, and at the end of the line the fold words and then the DERIVE words apply
    (register b reads as a + delta: its own stamps were dropped from the table, regex_handle.cpp planTdfaDerive)."""

    def __init__(self, rx, compact=True):
        super().__init__(rx, compact=compact)
        po = int(self.blob[7])
        assert po, "the blob carries no byte-pair extension"
        ph = [int(x) for x in self.blob[po // 4:po // 4 + 8]]
        self.p_base, self.p_row, off_cmapa, self.p_ida, fmt, off_derive = ph[:6]
        assert fmt == 1, "not a one-stamp pair table"
        self.cmapa = self.raw[off_cmapa:off_cmapa + 512].view(np.uint16)
        self.derive = []
        if off_derive:
            n = int(self.blob[off_derive // 4])
            self.derive = [(int(w) & 0xFF, (int(w) >> 8) & 0xFF, int(w) >> 16) for w in self.blob[off_derive // 4 + 1:off_derive // 4 + 1 + n]]
        self.doubles = 0

    @_with_run_captures
    def fullmatch_pair1(self, s: bytes, head=0, chunk=8):
        assert self.start_after is None or True
        L = len(s)
        dummy = self.nregs - 1
        regs = [0] * self.nregs
        state_row = self.p_base + (self.start_row - 320) // self.row_bytes * self.p_row
        id_col = self.id_col                                  # byte offset of the identity column in a single-byte row
        total = head + L if L else 0
        self.doubles = 0
        m = 0
        while m * chunk < total:
            pending = []                                      # DOUBLE entries of this chunk: (rB, value)
            for p in range(chunk // 2):
                i0 = m * chunk + 2 * p
                inside0, inside1 = head <= i0 < total, head <= i0 + 1 < total
                ca = int(self.cmapa[s[i0 - head]]) if inside0 else self.p_ida
                cb = int(self.cmap[s[i0 + 1 - head]]) if inside1 else id_col
                e = int(self.blob[(state_row + ca + cb) // 4])
                state_row = e & 0xFFFF
                pbase = i0 - head                             # line offset of the pair's first byte
                ra, delta = (e >> 16) & 0x7F, (e >> 23) & 1
                if ra != dummy:
                    assert 0 <= pbase + delta < L
                    regs[ra] = (pbase + delta) & (0xFFFF if self.compact else 0xFFFFFFFF)
                if e >> 31:
                    pending.append(((e >> 24) & 0x7F, (pbase + 1) & (0xFFFF if self.compact else 0xFFFFFFFF)))
            for rb, val in pending:                           # behind the chunk's rA stamps
                regs[rb] = max(regs[rb], val)
                self.doubles += 1
            m += 1
        for w in self.fold or ():
            for k in (1, 2, 3):
                r = (w >> (8 * k)) & 0xFF
                if r != 0xFF:
                    regs[r] = max(regs[r], regs[w & 0xFF])
        for b, a, delta in self.derive:
            regs[b] = (regs[a] + delta) & (0xFFFF if self.compact else 0xFFFFFFFF)
        state = (state_row - self.p_base) // self.p_row
        fid = int(self.final_id[state])
        if state == 0 or fid == 0xFFFF:
            return None
        caps = []
        for sl in range(self.nslots):
            mm = int(self.final_map[fid * self.nslots + sl])
            caps.append(L if mm == 0xFF else -1 if mm == 0xFE else regs[mm])
        return caps


class TdfaL2BlobInterp:
    """Walks the blob of the global-memory TDFA kernel (csrc/tdfa_l2_layout.h) exactly as tdfa_l2_kernel does: automata too
    large for the LDS kernels."""

    MISS = "miss"   # lazy automata: the value stepped on a transition nobody has computed (the kernels leave it LC_PENDING)

    def __init__(self, rx, which=None):
        blob = rx.table(B.LC_TABLE_TDFA_L2_BLOB if which is None else which, np.uint32)
        assert blob is not None
        raw = blob.view(np.uint8)
        (magic, self.nstates, self.ncls, self.nregs, self.nslots, self.start, o_trans, o_opsstart, o_ops, o_finalid, o_finalmap,
         o_after, total) = [int(x) for x in blob[:13]]
        assert magic == 0x324C4454 and total == blob.nbytes
        self.cmap = raw[64:320]
        self.trans = blob[o_trans // 4:o_trans // 4 + self.nstates * self.ncls]
        self.ops_start = blob[o_opsstart // 4:]
        self.ops = raw[o_ops:o_ops + (o_finalid - o_ops) // 2 * 2].view(np.uint16)
        self.final_id = raw[o_finalid:o_finalid + 2 * self.nstates].view(np.uint16)
        self.final_map = raw[o_finalmap:]
        self.start_after = blob[o_after // 4:o_after // 4 + self.ncls] if o_after else None
        self.runs = rx.run_captures()
        self.compact = False
        self.absorb = int(blob[13])                                                     # TL_ABSORB
        self.miss = int(blob[15])                                                       # TL_MISS (0: a complete automaton)
        o_quiet = int(blob[14])                                                         # TL_OFF_QUIET: u64[nStates]
        self.quiet = raw[o_quiet:o_quiet + 8 * self.nstates].view(np.uint64)

    @_with_run_captures
    def fullmatch_wave(self, s: bytes, start=0):
        """tdfa_wave_kernel's walk: stops in the absorbing state; a byte whose transition is "stay, no program" starts a scan for
        the first byte of a class outside the state's QUIET mask (the kernel does that 256 bytes at a time)."""
        state = self.start if start == 0 else int(self.start_after[int(self.cmap[s[start - 1]])])
        regs = [-1] * max(self.nregs, 1)
        pos, n = start, len(s)
        while pos < n and state != 0 and state != self.absorb and state != self.miss:
            t = int(self.trans[state * self.ncls + int(self.cmap[s[pos]])])
            if t >> 16:
                at = int(self.ops_start[t >> 16])
                for w in self.ops[at + 1:at + 1 + int(self.ops[at])]:
                    regs[int(w) & 0xFF] = pos if int(w) >> 8 == 0xFF else regs[int(w) >> 8]
            elif (t & 0xFFFF) == state:
                q = int(self.quiet[state])
                pos += 1
                while pos < n and int(self.cmap[s[pos]]) < 64 and (q >> int(self.cmap[s[pos]])) & 1:
                    pos += 1
                continue
            state = t & 0xFFFF
            pos += 1
        if state == 0:
            return None
        if self.miss and state == self.miss:
            return self.MISS
        fid = int(self.final_id[state])
        if fid == 0xFFFF:
            return None
        out = []
        for sl in range(self.nslots):
            m = int(self.final_map[fid * self.nslots + sl])
            out.append(len(s) if m == 0xFF else (-1 if m == 0xFE else regs[m]))
        return out

    @_with_run_captures
    def fullmatch(self, s: bytes, start=0):
        state = self.start if start == 0 else int(self.start_after[int(self.cmap[s[start - 1]])])
        regs = [-1] * max(self.nregs, 1)
        for pos in range(start, len(s)):
            t = int(self.trans[state * self.ncls + int(self.cmap[s[pos]])])
            if t >> 16:
                at = int(self.ops_start[t >> 16])
                for w in self.ops[at + 1:at + 1 + int(self.ops[at])]:
                    regs[int(w) & 0xFF] = pos if int(w) >> 8 == 0xFF else regs[int(w) >> 8]
            state = t & 0xFFFF
            if state == 0:
                return None
            if self.miss and state == self.miss:
                return self.MISS
        fid = int(self.final_id[state])
        if fid == 0xFFFF:
            return None
        out = []
        for sl in range(self.nslots):
            m = int(self.final_map[fid * self.nslots + sl])
            out.append(len(s) if m == 0xFF else (-1 if m == 0xFE else regs[m]))
        return out


def packed_tdfa_interp(rx):
    """the interpreter of whichever packed TDFA tables the handle carries: LDS kernels, or the global-memory kernel"""
    return TdfaBlobInterp(rx) if rx.table(B.LC_TABLE_TDFA_BLOB, np.uint32) is not None else TdfaL2BlobInterp(rx)
