"""debug: the late-rank / gone-peer scenario of tests/test_data_parallel_gpu.py step by step with progress lines"""
import ctypes, os, sys, time, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(90, exit=True)
import torch
from deepcv_amd import _native as nat
lib = nat.lib
device = torch.device('cuda:0')
cap, n = 1 << 15, 17010
nbytes = int(lib.dcv_allreduce_inbox_bytes(2, cap))
boxes = (ctypes.c_void_p * 2)()
for k in range(2):
    p = ctypes.c_void_p(); nat.check(lib.dcv_allreduce_alloc(nbytes, ctypes.byref(p)), 'alloc'); boxes[k] = p.value
print('alloc ok', flush=True)
words = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(2)]
for k in range(2):
    nat.check(lib.dcv_allreduce_set_error_word(boxes[k], words[k].data_ptr()), 'set_error_word')
print('error words ok', flush=True)
streams = [torch.cuda.Stream(device=device) for _ in range(2)]
g = torch.Generator().manual_seed(11)
a, b = torch.randn(n, generator=g).to(device), torch.randn(n, generator=g).to(device)
ra, rb = a.clone(), b.clone()
torch.cuda.synchronize()
nat.check(lib.dcv_allreduce_oneshot(ra.data_ptr(), n, 0, 2, boxes, cap, streams[0].cuda_stream), 'r0')
print('rank 0 launched', flush=True)
time.sleep(0.35)
print('rank 0 done?', streams[0].query(), flush=True)
nat.check(lib.dcv_allreduce_oneshot(rb.data_ptr(), n, 1, 2, boxes, cap, streams[1].cuda_stream), 'r1')
print('rank 1 launched', flush=True)
t0 = time.time()
while not (streams[0].query() and streams[1].query()):
    if time.time() - t0 > 20: print('STUCK: r0 done', streams[0].query(), 'r1 done', streams[1].query(), flush=True); os._exit(3)
    time.sleep(0.01)
print('both done in', time.time() - t0, 'equal', torch.equal(ra, (a + b) * 0.5), torch.equal(rb, (a + b) * 0.5), flush=True)
os.environ['DCV_ALLREDUCE_TIMEOUT_S'] = '0.5'
ra = a.clone(); t0 = time.time()
nat.check(lib.dcv_allreduce_oneshot(ra.data_ptr(), n, 0, 2, boxes, cap, streams[0].cuda_stream), 'r0 alone')
while not streams[0].query():
    if time.time() - t0 > 20: print('STUCK alone', flush=True); os._exit(4)
    time.sleep(0.01)
print('alone done in', time.time() - t0, 'all nan', bool(torch.isnan(ra).all()), 'words', int(words[0][0]), int(words[1][0]), flush=True)
