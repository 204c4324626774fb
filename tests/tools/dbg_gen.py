import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from oracle import bn254 as O, cpu_ref as Cr
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
def gen(t0, d, n):
    out = torch.zeros(n * 8, dtype=torch.int64, device="cuda"); torch.cuda.synchronize()
    a, b = F.fr_encode([t0])[0], F.fr_encode([d])[0]
    _lib.check(lib.zkhip_g1_gen_walk_device(a.ctypes.data, b.ctypes.data, n, out.data_ptr(), None))
    torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint64).reshape(n, 8)
for (t0, d, n) in [] if os.environ.get('DBG_SIZES') else [(1, 0, 1), (2, 0, 1), (1, 1, 4), (3, 5, 40), (0x1234567, 0xABCDEF0123456789ABCDEF, 3), (0x1234567, 1, 3), (1<<200, 1, 2)]:
    got = gen(t0, d, n)
    for i in range(min(n, 4)) if n < 40 else (0, 1, 31, 32, 39):
        exp = O.scalar_mul((t0 + i * d) % O.R_MOD, O.G1_GEN)
        g = O.affine_from_limbs([int(x) for x in got[i]])
        # find which multiple it is for small cases
        which = None
        if g is not None and O.on_curve(g):
            for k in range(0, 300):
                if O.scalar_mul(k, O.G1_GEN) == g: which = k; break
        print(f"t0={t0:#x} d={d:#x} i={i}: ok={g == exp} on_curve={g is None or O.on_curve(g)} small_multiple={which}")

# phase profile at 2^22 / 2^24 for both paths
def profile_read():
    ms = (C.c_double * 32)(); names = ((C.c_char * 64) * 32)()
    k = lib.zkhip_profile_read(ms, names, 32)
    return {names[i].value.decode(): round(ms[i], 3) for i in range(max(k, 0))}
stream = torch.cuda.current_stream().cuda_stream
for L in ([int(x) for x in os.environ['DBG_SIZES'].split(',')] if os.environ.get('DBG_SIZES') else (20, 22)):
    n = 1 << L
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    a77, b991 = F.fr_encode([77])[0], F.fr_encode([991])[0]
    _lib.check(lib.zkhip_g1_gen_walk_device(a77.ctypes.data, b991.ctypes.data, n, bases.data_ptr(), stream))
    rng = np.random.default_rng(L)
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); a[:, 3] = rng.integers(0, 0x30644E72E131A029, size=n, dtype=np.uint64)
    sc = torch.from_numpy(a.view(np.int64)).cuda()
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
    lib.zkhip_profile_enable(1)
    for rep in range(2):
        _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, sc.data_ptr(), n, out.data_ptr(), stream)); p1 = profile_read()
        _lib.check(lib.zkhip_msm_g1_device(sc.data_ptr(), bases.data_ptr(), n, out.data_ptr(), stream)); p2 = profile_read()
    print(f"2^{L} prepared: total {sum(p1.values()):.3f}", p1)
    print(f"2^{L} general : total {sum(p2.values()):.3f}", p2)
    if L == 22:
        # tiling hypothesis: 4 sub-range calls of 2^20
        tot = 0
        for q in range(4):
            _lib.check(lib.zkhip_msm_g1_prepared_device(h, q << 20, sc.data_ptr() + (q << 20) * 32, 1 << 20, out.data_ptr(), stream)); pq = profile_read()
            tot += sum(pq.values()); print("   tile", q, pq.get("accumulate"))
        print(f"2^22 as 4 tiles of 2^20 against the 2^22 table: {tot:.3f} ms")
    lib.zkhip_profile_enable(0)
    lib.zkhip_release_bases(h)
