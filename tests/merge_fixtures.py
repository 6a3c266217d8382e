"""Hand-written groups for the merge processor (ProcessorMergeMultilineLogNative) and what it must leave of them.  Two readers: the GPU
tests of the product (tests/test_multiline.py) and the CPU test that runs the REFERENCE's own merge processor on the same groups
(tests/test_reference_neighbours.py), so the expectations the product is held to are what the reference's code does."""


def _ev(text, ts, part=False):
    contents = [["content", text]] if text is not None else []
    if part:
        contents.append(["P", ""])
    return {"contents": contents, "timestamp": ts, "type": 1}


def flag_group(with_meta):
    """MergeLogsByFlag :113-159: runs of events that carry the "P" content are joined, without line feeds, with the first event behind
    them that has no flag; a run at the end of the group is joined as it is"""
    g = {"events": [_ev("aaa", 1, True), _ev("bbb", 2, True), _ev("ccc", 3), _ev("single", 4), _ev("tail1", 5, True), _ev("tail2", 6, True)]}
    if with_meta:
        g["metadata"] = {"has.part.log": "P"}
    return g


FLAG_TIMESTAMPS = [1, 4, 5]                       # the first event of every merged run survives
FLAG_CONTENTS = ["aaabbbccc", "single", "tail1tail2"]
FLAG_COUNTERS = (6, 0)                            # (merged events, unmatched events)

# HandleUnmatchLogs counts and moves EVENTS [begin, cur] (:360-392): events without any content, which the walk skips (:190), are inside
# such a range when a log fails its end pattern, and behind the last item at the flush.
# (events, config, timestamps left, (merged, unmatched))
EMPTY_EVENT_CASES = [
    # continue + end: " a", (empty), " b", "x" -- "x" is neither continuation nor end: the three events before it and "x" go to
    # HandleUnmatchLogs as ONE range of 4 events; then the flush [5..6] with the empty event behind the last item
    ([(" a", 1), (None, 2), (" b", 3), ("x", 4), (" c", 5), (None, 6)],
     {"ContinuePattern": r"\s+.*", "EndPattern": "END"}, [1, 2, 3, 4, 5, 6], (0, 6)),
    # discard: the same events are counted and dropped
    ([(" a", 1), (None, 2), (" b", 3), ("x", 4), (" c", 5), (None, 6)],
     {"ContinuePattern": r"\s+.*", "EndPattern": "END", "UnmatchedContentTreatment": "discard"}, [], (0, 6)),
    # only an end pattern: the last item closes a log, empty events behind it still reach HandleUnmatchLogs (:316)
    ([("END", 1), (None, 2)], {"EndPattern": "END"}, [1, 2], (1, 1)),
]


def empty_event_group(events):
    return {"events": [_ev(text, ts) for text, ts in events]}
