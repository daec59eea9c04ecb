"""What the HIP runtime charges for the calls a plan makes when it is built and destroyed (ctypes on libamdhip64): stream
create / destroy (3.7 / 2.5 ms each on ROCm 7.2 -- why libnmx recycles streams), event create, malloc / memset / free."""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
hip.hipSetDevice(0)
hip.hipFree(None)
s = C.c_void_p()
ts = []
streams = []
for i in range(24):
    t0 = time.perf_counter(); hip.hipStreamCreateWithFlags(C.byref(s), 1); ts.append(time.perf_counter() - t0); streams.append(C.c_void_p(s.value))
print("create us", [round(t * 1e6) for t in ts])
td = []
for st in streams:
    t0 = time.perf_counter(); hip.hipStreamDestroy(st); td.append(time.perf_counter() - t0)
print("destroy us", [round(t * 1e6) for t in td])
e = C.c_void_p(); te = []
for i in range(10):
    t0 = time.perf_counter(); hip.hipEventCreateWithFlags(C.byref(e), 2); te.append(time.perf_counter() - t0)
print("event create us", [round(t * 1e6) for t in te])
p = C.c_void_p(); tm = []
for n in (1 << 20, 1 << 24, 1 << 26):
    t0 = time.perf_counter(); hip.hipMalloc(C.byref(p), C.c_size_t(n)); t1 = time.perf_counter(); hip.hipMemset(p, 0, C.c_size_t(n)); hip.hipDeviceSynchronize(); t2 = time.perf_counter(); hip.hipFree(p); t3 = time.perf_counter()
    print(n, "malloc us", round((t1 - t0) * 1e6), "memset+sync us", round((t2 - t1) * 1e6), "free us", round((t3 - t2) * 1e6))
