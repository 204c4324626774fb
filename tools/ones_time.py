import os, sys, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
L = 22; n = 1 << L
t0m, dm = F.fr_encode([77])[0], F.fr_encode([991])[0]
bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
_lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
h = C.c_uint64(0)
_lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
one = torch.from_numpy(F.fr_encode([1])[0].view(np.int64)).cuda()
out = torch.zeros(12, dtype=torch.int64, device="cuda")
def timed(x, reps=10):
    f = lambda: _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, x.data_ptr(), n, out.data_ptr(), None))
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
def phases(x):
    lib.zkhip_profile_enable(1)
    _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, x.data_ptr(), n, out.data_ptr(), None))
    ms = (C.c_double * 32)(); names = ((C.c_char * 64) * 32)()
    k = lib.zkhip_profile_read(ms, names, 32)
    lib.zkhip_profile_enable(0)
    return " ".join(f"{names[i].value.decode()}={ms[i]:.3f}" for i in range(k))
x = one.repeat(n, 1).contiguous()
print("all ones 2^22: %.3f ms" % timed(x), phases(x))
sel = torch.zeros((n, 4), dtype=torch.int64, device="cuda"); sel[::2] = one
print("selector (half ones) 2^22: %.3f ms" % timed(sel), phases(sel))
r = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda"); r[:, 3] &= (1 << 61) - 1
r[::20] = one
print("uniform with 5%% ones 2^22: %.3f ms" % timed(r), phases(r))
r[::20] = torch.randint(0, 1 << 62, (n // 20 + (1 if n % 20 else 0), 4), dtype=torch.int64, device="cuda") & ((1 << 61) - 1)
print("uniform 2^22: %.3f ms" % timed(r), phases(r))
