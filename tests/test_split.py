"""Line splitting: the split oracle against hand-checked reference behaviours on the CPU; the
device kernels against the oracle, and split -> match with no host round trip, on the GPU."""
import numpy as np
import pytest

from oracle.split_oracle import split_lines, split_table


def test_split_oracle_reference_behaviours():
    # ProcessorSplitLogStringNative.cpp:131-174
    assert split_lines(b"") == []
    assert split_lines(b"abc") == [(0, 3)]                      # unterminated tail is a line
    assert split_lines(b"abc\n") == [(0, 3)]                    # trailing separator opens no new line
    assert split_lines(b"a\n\nb") == [(0, 1), (2, 0), (3, 1)]   # empty lines are lines
    assert split_lines(b"\n") == [(0, 0)]
    assert split_lines(b"\n\n") == [(0, 0), (1, 0)]
    assert split_lines(b"a\r\nb") == [(0, 2), (3, 1)]           # \r is payload
    assert split_lines(b"a|b|", ord("|")) == [(0, 1), (2, 1)]
    off = split_table(b"ab\ncd")
    assert list(off) == [0, 3, 6]                               # len = off[i+1]-off[i]-1 -> 2, 2


@pytest.mark.gpu
def test_split_kernels_match_the_oracle():
    import torch
    from loongcollector_amd import binding as B
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(9)
    cases = [b"", b"x", b"\n", b"abc", b"abc\n", b"a\n\nb", b"\n\n\n", b"a" * 100, b"a\n" * 40000 + b"tail"]
    for n in (1, 15, 16, 17, 63, 64, 65, 4097, 16384, 16385, 70000, 300001):
        arr = rng.choice(np.frombuffer(b"ab \n\n", dtype=np.uint8), size=n)
        cases.append(arr.tobytes())
    big = rng.choice(np.frombuffer(b"abcdefg\n", dtype=np.uint8), size=5_000_003)
    cases.append(big.tobytes())
    for split_char in (10, ord("a"), 0):
        for buf in cases:
            if split_char == 0:
                buf = buf.replace(b"b", b"\x00")
            exp = split_table(buf, split_char)
            nbytes = len(buf)
            cap = len(exp) + 8
            d_data = torch.from_numpy(np.frombuffer(buf + b"\x00" * 16, dtype=np.uint8).copy()).to(dev)
            d_off = torch.full((cap,), 0x7FFFFFFF, dtype=torch.int32, device=dev)
            d_n = torch.full((1,), -1, dtype=torch.int32, device=dev)
            d_scratch = torch.empty((B.split_scratch_bytes(nbytes) // 4 + 1,), dtype=torch.int32, device=dev)
            B.split_lines_device(d_data, nbytes, d_off, d_n, d_scratch, split_char=split_char,
                                 stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            n = int(d_n.item())
            assert n == len(exp) - 1 if nbytes else n == 0, (split_char, nbytes, n)
            if nbytes:
                got = d_off.cpu().numpy().view(np.uint32)[:n + 1]
                assert np.array_equal(got, exp), (split_char, nbytes)


@pytest.mark.gpu
def test_split_then_match_without_host_round_trip():
    """raw read buffer -> split kernels -> match kernel on one stream; result equals the oracle's split + match."""
    import torch
    from loongcollector_amd import binding as B
    from loongcollector_amd import corpus
    from oracle.oracle import OracleRegex
    dev = torch.device("cuda:0")
    n = 50000
    data, off, length = corpus.apache_batch(n, "A", poison_every=37)
    raw = data.tobytes()[:-1]                     # last line unterminated, like the tail of a read buffer
    lines = split_lines(raw)
    assert len(lines) == n
    rx = B.GpuRegex(corpus.REGEX_A)
    G = rx.groups
    cap = n + 100
    d_data = torch.from_numpy(np.frombuffer(raw + b"\x00" * 16, dtype=np.uint8).copy()).to(dev)
    d_off = torch.zeros((cap + 1,), dtype=torch.int32, device=dev)
    d_n = torch.zeros((1,), dtype=torch.int32, device=dev)
    d_scratch = torch.empty((B.split_scratch_bytes(len(raw)) // 4 + 1,), dtype=torch.int32, device=dev)
    d_caps = torch.full((cap, 2 * G), -9, dtype=torch.int32, device=dev)
    d_status = torch.full((cap,), 7, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    B.split_lines_device(d_data, len(raw), d_off, d_n, d_scratch, stream=s)
    rx.match_device_dyn(d_data, d_off, d_n, cap, d_caps, d_status, stream=s)
    torch.cuda.synchronize()
    assert int(d_n.item()) == n
    exp_caps, exp_status = OracleRegex(corpus.REGEX_A).fullmatch_batch(
        np.frombuffer(raw, dtype=np.uint8), np.array([b for b, _ in lines], np.uint32), np.array([l for _, l in lines], np.uint32))
    assert np.array_equal(d_status.cpu().numpy()[:n], exp_status)
    assert np.array_equal(d_caps.cpu().numpy()[:n], exp_caps)
    assert (d_status.cpu().numpy()[n:] == 7).all()      # lines beyond *d_nlines are not touched
