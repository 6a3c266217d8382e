"""oracle/split_oracle.py -- TEST INFRASTRUCTURE ONLY.

Restates ProcessorSplitLogStringNative::ProcessEvent's line walk + GetNextLine
(core/plugin/processor/inner/ProcessorSplitLogStringNative.cpp:130-174): lines are the SplitChar-delimited segments of
the buffer; an unterminated tail is a line; empty segments are lines; a trailing SplitChar opens no new line."""
import numpy as np


def split_lines(buf: bytes, split_char: int = 10):
    """-> list of (begin, length), exactly the StringViews the reference creates"""
    out = []
    begin = 0
    n = len(buf)
    while begin < n:                       # :131
        end = begin
        while end < n and buf[end] != split_char:   # GetNextLine :164-173
            end += 1
        out.append((begin, end - begin))
        begin += (end - begin) + 1         # :160
    return out


def split_table(buf: bytes, split_char: int = 10):
    """The device form: off[n+1] with len[i] = off[i+1]-off[i]-1."""
    lines = split_lines(buf, split_char)
    off = np.zeros(len(lines) + 1, dtype=np.uint32)
    for i, (b, l) in enumerate(lines):
        off[i] = b
        off[i + 1] = b + l + 1
    return off
