import ctypes, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from loongcollector_amd import binding, corpus
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
n = 1 << 20
data, off, length = corpus.apache_batch(n, "A")
rx = binding.GpuRegex(corpus.REGEX_A); G = rx.groups
d_data = torch.from_numpy(data).to(dev); d_off = torch.from_numpy(off.view(np.int32)).to(dev)
d_caps = torch.empty((n, 2*G), dtype=torch.int32, device=dev); d_status = torch.empty((n,), dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream()
L = binding.load()
args = (rx.handle, ctypes.c_int(0), ctypes.c_void_p(d_data.data_ptr()), ctypes.c_void_p(d_off.data_ptr()), ctypes.c_void_p(None),
        ctypes.c_uint32(1), ctypes.c_uint32(n), ctypes.c_uint32(G), ctypes.c_void_p(d_caps.data_ptr()), ctypes.c_void_p(d_status.data_ptr()), ctypes.c_void_p(stream.cuda_stream))
fn = L.lc_regex_match_device_engine
for _ in range(3): fn(*args)
torch.cuda.synchronize()
for K in (20, 100):
    t0 = time.perf_counter()
    for _ in range(K): fn(*args)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("K=%d launch-only %.3f ms/step, with sync %.3f ms/step" % (K, (t1-t0)/K*1e3, (t2-t0)/K*1e3))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(200)]
t0 = time.perf_counter()
for e in ev: e.record(stream)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("event.record %.3f ms each" % ((t1-t0)/200*1e3))
t0 = time.perf_counter()
for _ in range(1000): L.lc_device_count()
print("lc_device_count %.4f ms" % ((time.perf_counter()-t0)))
# default (null) stream
args0 = args[:-1] + (ctypes.c_void_p(None),)
t0 = time.perf_counter()
for _ in range(100): fn(*args0)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("null stream: launch-only %.3f ms/step, with sync %.3f ms/step" % ((t1-t0)/100*1e3, (t2-t0)/100*1e3))
