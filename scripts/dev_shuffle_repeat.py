import sys, torch, time
sys.path.insert(0, '/root/repo')
from biomedkg_amd import rand as R
dev='cuda'
et = torch.randint(0, 30, (4_000_000,), device=dev)
r = R.DeviceRand(seed=1)
for k in (1, 10):
    for _ in range(3): r.shuffle_repeat(et, k)
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): r.shuffle_repeat(et, k)
    b.record(); torch.cuda.synchronize()
    print("shuffle_repeat k=%d: %.3f ms" % (k, a.elapsed_time(b)/10))
    x = et.repeat(k)
    for _ in range(3): r.shuffle(x)
    a.record()
    for _ in range(10): r.shuffle(x)
    b.record(); torch.cuda.synchronize()
    print("shuffle(repeat) k=%d: %.3f ms" % (k, a.elapsed_time(b)/10))
