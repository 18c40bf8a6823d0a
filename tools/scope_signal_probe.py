"""Where does a run under ROC_SYSTEM_SCOPE_SIGNAL=0 stop?  (gpurun_out/ab_env.txt, round 1: bench.py hung under that setting.)
Stages print before / after; run under `timeout`:  ROC_SYSTEM_SCOPE_SIGNAL=0 timeout -k 5 90 python tools/scope_signal_probe.py"""
import faulthandler, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(40, exit=True)
def say(*a): print(*a, flush=True)
say('env ROC_SYSTEM_SCOPE_SIGNAL =', os.environ.get('ROC_SYSTEM_SCOPE_SIGNAL'))
import torch
dev = torch.device('cuda:0')
say('stage 1: torch kernel + stream synchronize ...')
x = torch.ones(1 << 20, device=dev); y = (x * 2).sum(); torch.cuda.synchronize(); say('  ok', float(y))
say('stage 2: event wait ...')
e = torch.cuda.Event(); z = x + 1; e.record(); e.synchronize(); say('  ok')
say('stage 3: kernel writing pinned host memory behind __threadfence_system, host polls (no stream sync) ...')
import ctypes
from deepcv_amd import _native as nat
mb = torch.zeros(1024, dtype=torch.uint8).pin_memory(); view = mb.numpy()
import struct
val = torch.full((1,), 3.25, device=dev)
struct.pack_into('<8Q8Q8Qii', view, 0, *([0] * 24), 0, 4242)          # mailbox with no copy jobs, sequence number 4242
ptrs = (ctypes.c_void_p * 1)(val.data_ptr())
nat.check(nat.lib.dcv_publish_scalars(mb.data_ptr(), ptrs, 1, mb.data_ptr() + 768, mb.data_ptr() + 832, nat.current_stream(dev)))
t0 = time.monotonic()
while int(view[832:836].view('<i4')[0]) != 4242 and time.monotonic() - t0 < 10: pass
say('  sequence number seen after %.1f ms:' % ((time.monotonic() - t0) * 1e3), int(view[832:836].view('<i4')[0]), 'value', float(view[768:772].view('<f4')[0]))
torch.cuda.synchronize(); say('  after stream sync:', int(view[832:836].view('<i4')[0]))
say('stage 4: hipGraph capture + replay + synchronize ...')
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
with torch.cuda.graph(g, stream=s): w = x * 3
g.replay(); torch.cuda.synchronize(); say('  ok')
say('done')
