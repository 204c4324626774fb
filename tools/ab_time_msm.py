#!/usr/bin/env python3
"""A/B helper: median / min time of the prepared MSM (2^L) over many runs, and the median of every phase (in-library HIP events).
  ab_time_msm.py [log_n=20] [reps=100] [label]      knobs: ZKHIP_MAX_WINDOW (cap the prepared window), ZKHIP_TASK_SHIFT (task length 2^s),
  ZKHIP_MSM_PIECES (bucket-range pieces of the wide path), ZKHIP_NO_OVERLAP=1 (no second stream: the pieces run one after the other)
Run the variants alternately in one gpurun call: box-to-box differences (+-4 %) exceed most single-kernel effects."""
import os, sys, ctypes as C, statistics as st
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n = 1 << L
stream = torch.cuda.current_stream().cuda_stream
t0, dd = F.fr_encode([77])[0], F.fr_encode([991])[0]
bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
_lib.check(lib.zkhip_g1_gen_walk_device(t0.ctypes.data, dd.ctypes.data, n, bases.data_ptr(), stream))
rng = np.random.default_rng(1)
a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); a[:, 3] = rng.integers(0, 0x30644E72E131A029, size=n, dtype=np.uint64)
sc = torch.from_numpy(a.view(np.int64)).cuda()
out = torch.zeros(12, dtype=torch.int64, device="cuda")
h = C.c_uint64(0)
_lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
def run(): _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, sc.data_ptr(), n, out.data_ptr(), stream))
for _ in range(10): run()
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
ms = (C.c_double * 32)(); names = ((C.c_char * 64) * 32)()
acc = {}
lib.zkhip_profile_enable(1)
for _ in range(30):
    run(); torch.cuda.synchronize()
    k = lib.zkhip_profile_read(ms, names, 32)
    per = {}
    for i in range(k): per[names[i].value.decode()] = per.get(names[i].value.decode(), 0.0) + ms[i]     # a phase may be marked once per bucket-range piece
    for nm, v in per.items(): acc.setdefault(nm, []).append(v)
lib.zkhip_profile_enable(0)
print(f"{sys.argv[3] if len(sys.argv) > 3 else '':8s} msm median {st.median(ts):.4f} min {min(ts):.4f} | " + " ".join(f"{k} {st.median(v):.4f}" for k, v in acc.items()))
