"""Random shapes through the bf16 convolution entry points, checked like tests/test_bf16_gpu.py's fixed cases (same functions, same bounds against torch in double
precision): every kernel family of conv2d_bf16.hip is chosen by shape, so random extents walk through the dispatch boundaries.  usage: bf16_fuzz.py [cases] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_bf16_gpu as T
from deepcv_amd import _native as nat
dev = torch.device('cuda:0')
count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
bad = 0
t0 = time.time()
for i in range(count):
    cin, cout = rng.choice([64, 64, 128, 192, 256]), rng.choice([64, 128, 192, 256])
    k, stride, pad, dil = rng.choice([(3, 1, 1, 1)] * 5 + [(3, 2, 1, 1), (1, 1, 0, 1), (1, 2, 0, 1), (5, 1, 2, 1), (3, 1, 2, 2), (3, 2, 1, 1)])
    h, w = rng.randint(4, 40), rng.randint(4, 40)
    if rng.random() < 0.3: w = rng.choice([4, 8, 16, 32])
    budget = 6e6 / (cin * max(h * w, 1))                       # keep the CPU reference in double precision short
    n = max(1, min(int(budget), rng.choice([1, 2, 3, 7, 20, 70, 130])))
    act, has_bias = rng.choice([0, 1, 1, 2, 4]), rng.random() < 0.7
    case = (n, cin, h, w, cout, k, stride, pad, dil, act, has_bias)
    d = T._desc(nat, n, cin, h, w, cout, k, stride, pad, dil, act)
    if d.ho <= 0 or d.wo <= 0 or not nat.lib.dcv_conv2d_bf16_supported(d, 0): continue
    try:
        T.test_bf16_forward_and_statistics(dev, case, rng.random() < 0.5)
        T.test_bf16_data_and_weight_gradients(dev, case)
        print(f'ok   {case}  [{time.time() - t0:.0f} s]', flush=True)
    except AssertionError as e:
        bad += 1
        print(f'FAIL {case}: {e}', flush=True)
print(f'{bad} failures', flush=True)
sys.exit(1 if bad else 0)
