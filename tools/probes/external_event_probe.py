"""Does an EXTERNAL event recorded inside a captured hipGraph (hipEventRecordWithFlags(..., hipEventRecordExternal): an event-record node) order a wait issued on
another stream right after each replay?  torch refuses external events on ROCm, so the HIP runtime is called through ctypes.  Graph: a = x * k (then a chain of
kernels), [event], b = a + 1 (a longer chain).  After each replay a side stream waits for the event and copies `a`: it must hold THIS replay's values (k changes per
replay) and the copy must finish before the graph's tail does.  Decides how captured data-parallel steps can start a bucket's all-reduce inside the backward pass."""
import ctypes, time, torch
hip = ctypes.CDLL('libamdhip64.so')
dev = torch.device('cuda:0')
n = 1 << 24
x = torch.ones(n, device=dev); k = torch.ones((), device=dev); a = torch.empty_like(x); b = torch.empty_like(x); tail = torch.empty_like(x)
s = torch.cuda.Stream(); side = torch.cuda.Stream()
ev = ctypes.c_void_p()
assert hip.hipEventCreateWithFlags(ctypes.byref(ev), 2) == 0          # hipEventDisableTiming
g = torch.cuda.CUDAGraph()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    torch.mul(x, k, out=a)
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s):
    torch.mul(x, k, out=a)
    for _ in range(20): a.mul_(1.0)
    rc = hip.hipEventRecordWithFlags(ev, ctypes.c_void_p(s.cuda_stream), 1)      # hipEventRecordExternal
    torch.add(a, 1.0, out=b)
    for _ in range(200): b.mul_(1.0)
    torch.add(b, 0.0, out=tail)
print('hipEventRecordWithFlags(external) in capture ->', rc)
ok = rc == 0
for it in range(1, 6):
    k.fill_(float(it)); torch.cuda.synchronize()
    g.replay()
    rcw = hip.hipStreamWaitEvent(ctypes.c_void_p(side.cuda_stream), ev, 0)
    with torch.cuda.stream(side):
        got = a[:4].clone()
        done = torch.cuda.Event(); done.record(side)
    t0 = time.perf_counter()
    while not done.query():
        pass
    t_side = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    v = got.cpu().tolist()
    print(f'replay {it}: wait rc {rcw}, side copy saw {v[0]} (want {float(it)}), side done after {t_side * 1e3:.2f} ms, graph done after {t_all * 1e3:.2f} ms')
    ok = ok and rcw == 0 and v[0] == float(it) and t_side < 0.8 * t_all
print('EXTERNAL EVENT OK' if ok else 'EXTERNAL EVENT NOT USABLE')
