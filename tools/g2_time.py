import os, sys, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
from tests.test_gpu_g2 import enc, walk
lib = _lib.load()
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
pts = enc(walk(31337, 4242, 256))
pts = np.ascontiguousarray(np.tile(pts, (n // 256, 1)))
rng = np.random.default_rng(1)
a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); a[:, 3] = rng.integers(0, 0x30644E72E131A029, size=n, dtype=np.uint64)
dp = torch.from_numpy(pts.view(np.int64)).cuda(); ds = torch.from_numpy(a.view(np.int64)).cuda()
out = torch.zeros(24, dtype=torch.int64, device="cuda")
f = lambda: _lib.check(lib.zkhip_msm_g2_device(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr(), None))
f(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3): f()
torch.cuda.synchronize(); print("g2 n=%d %.3f ms" % (n, (time.perf_counter() - t) / 3 * 1e3))
lib.zkhip_profile_enable(1); f()
tm = (C.c_double * 32)(); names = ((C.c_char * 64) * 32)(); k = lib.zkhip_profile_read(tm, names, 32)
print(" ".join(f"{names[i].value.decode()}={tm[i]:.3f}" for i in range(k)))
