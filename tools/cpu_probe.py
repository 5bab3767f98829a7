import os, sys, time
sys.path.insert(0, 'tests')
import numpy as np, oracle_lib as o, ctypes as C
print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
img = o.pad_frame(o.gen_frame(3840, 2160, 0)); lut = o.lut_for(False, 5)
x = o.level_shift_fwd(img, False)
for n in (1, 8, 16, 32, 64):
    o.set_threads(n)
    f = o.dwt_forward(x, 5); coef = f[:img.size].reshape(img.shape)
    t = time.time(); f = o.dwt_forward(x, 5); t2 = time.time() - t
    t = time.time(); st, sz = o.bpc_encode(coef, 5, lut); t3 = time.time() - t
    print(n, 'threads dwt %.3f bpc %.3f -> %.1f Mpx/s' % (t2, t3, 3840 * 2160 / (t2 + t3) / 1e6))
