"""us per Normalizer.normalize() call (running-statistics push + normalisation of the batch): python tools/normalizer_timing.py"""
import torch, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from omnisafe_amd.normalizer import Normalizer
dev='cuda:0'
for N, D in ((4096, 60), (4096, 376), (16, 60), (65536, 60)):
    nm = Normalizer((D,), clip=5.0, device=dev)
    x = torch.randn(N, D, device=dev)
    for _ in range(3): nm.normalize(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): nm.normalize(x)
    b.record(); torch.cuda.synchronize()
    print(N, D, 'push+normalize us', round(a.elapsed_time(b) * 1e3 / 50, 2))
