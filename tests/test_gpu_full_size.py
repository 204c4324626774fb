"""GPU (-m gpu): parity evidence at the BASELINE sizes (configs[1], [3], [4]) -- the cases the small-size suite cannot reach.

* NTT on *random* data at 2^24 .. 2^28: NTT(a)[i] = a(omega^i) for sampled i, with a(x) evaluated by the oracle's restatement
  of halo2's `eval_polynomial` (one Horner pass over all n inputs per sample, so every input column and every output tile
  position takes part; a round trip or a single delta cannot see a consistent mis-placement).
* `coeff_to_extended` at the wrapper size (k = 22 -> 24): out[i] = a(zeta * omega_ext^i).
* MSM at 2^24 with uniform scalars (prepared path), the general path at 2^20, a ragged 2^21 + 5 prepared MSM (the per-GPU
  shard size of configs[4], where the wide sort switches kernels), all against the structured-SRS identity.
* workspace sizing at the task-count switch (n = 2^19 - 1, 2^19 - 8 on a fresh context; batched non-power-of-two sizes).
"""
import ctypes as C

import numpy as np
import pytest

from oracle import bn254 as O
from zksnap_circuits_halo2_amd import _lib, fields as F

pytestmark = pytest.mark.gpu


def _random_fr_device(torch, n, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    a = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    a[:, 3] = torch.randint(0, 1 << 61, (n,), dtype=torch.int64, device="cuda", generator=g)   # < r: canonical Montgomery words
    return a


def _samples(n):
    return [0, 1, 2, 2047, 2048, 4097, 65537, (n >> 1) + 3, n - (n >> 3) + 11, n - 1]


def _eval_at(cref, coeffs_dev, n, points):
    """[a(x) for x in points] for n device-resident coefficients: the oracle's `eval_polynomial` on 2^24-element slices (512 MiB of
    host memory whatever n is), slice values scaled by x^start and summed with Python integers"""
    threads = cref.usable_cores()
    step = min(n, 1 << 24)
    acc = [0] * len(points)
    for lo in range(0, n, step):
        host = np.ascontiguousarray(coeffs_dev[lo:lo + step].cpu().numpy().view(np.uint64))
        for t, x in enumerate(points):
            part = F.fr_decode(cref.eval_polynomial_mt(host, F.fr_encode([x])[0], threads))[0]
            acc[t] = (acc[t] + part * pow(x, lo, O.R_MOD)) % O.R_MOD
    return F.fr_encode(acc)


@pytest.mark.parametrize("log_n", [24, 26, 28])
def test_ntt_random_data_equals_polynomial_evaluation(lib, cref, log_n):
    import torch

    n = 1 << log_n
    w = O.omega_for(log_n)
    om = F.fr_encode([w])[0]
    a = _random_fr_device(torch, n, 2400 + log_n)
    idx = _samples(n)
    exp = _eval_at(cref, a, n, [pow(w, i, O.R_MOD) for i in idx])   # from the coefficients, before the in-place transform
    _lib.check(lib.zkhip_ntt_fr_device(a.data_ptr(), om.ctypes.data, log_n, None))
    torch.cuda.synchronize()
    got = a[idx].cpu().numpy().view(np.uint64)
    assert np.array_equal(got, exp), (log_n, [i for i, g, e in zip(idx, got, exp) if not np.array_equal(g, e)])


def test_coeff_to_extended_wrapper_size_equals_coset_evaluation(lib, cref):
    """k = 22 -> extended k = 24 (the wrapper circuit's quotient domain), device-resident, batch of 2 with a padded stride."""
    import torch

    k, ek = 22, 24
    n, en = 1 << k, 1 << ek
    w = O.omega_for(ek)
    zeta = F.ZETA
    a = _random_fr_device(torch, 2 * n + 64, 2224)               # polynomial b at b * (n + 64)
    out = torch.zeros((2 * en, 4), dtype=torch.int64, device="cuda")
    om, z = F.fr_encode([w])[0], F.fr_encode([zeta])[0]
    _lib.check(lib.zkhip_coeff_to_extended_device(a.data_ptr(), n + 64, k, out.data_ptr(), en, ek, 2, om.ctypes.data, z.ctypes.data, None))
    torch.cuda.synchronize()
    for b in range(2):
        idx = _samples(en)
        exp = _eval_at(cref, a[b * (n + 64): b * (n + 64) + n], n, [zeta * pow(w, i, O.R_MOD) % O.R_MOD for i in idx])
        got = out[[b * en + i for i in idx]].cpu().numpy().view(np.uint64)
        assert np.array_equal(got, exp), (b, [i for i, g, e in zip(idx, got, exp) if not np.array_equal(g, e)])
    # and back: extended_to_coeff(coeff_to_extended(p)) = p, zero above n
    omi, div = F.fr_encode([pow(w, -1, O.R_MOD)])[0], F.fr_encode([pow(en, -1, O.R_MOD)])[0]
    back = torch.zeros((3 * n, 4), dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_extended_to_coeff_device(out.data_ptr(), en, ek, omi.ctypes.data, div.ctypes.data, z.ctypes.data, back.data_ptr(), 3 * n, 3 * n, 1, None))
    torch.cuda.synchronize()
    assert torch.equal(back[:n], a[:n]) and not bool(back[n:].any())


def _walk_bases(lib, torch, n, T0, D):
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    t0m, dm = F.fr_encode([T0])[0], F.fr_encode([D])[0]          # named: `.ctypes.data` of a temporary would dangle
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
    return bases


def _expect(cref, sc, T0, D):
    return cref.jac_to_affine(cref.scalar_mul(cref.expected_scalar(sc, T0, D), cref.generator()))


def _aff(cref, t):
    return cref.jac_to_affine(np.ascontiguousarray(t.cpu().numpy().view(np.uint64)[:12]))


def test_msm_2p24_uniform_scalars_prepared(lib, cref):
    """configs[4]'s total size with full-weight scalars (the witness-like case does a third of the work)."""
    import torch

    n = 1 << 24
    T0, D = 0x5A4B534E41500002 + 240, 0x9E3779B97F4A7C15F39CC0605CEDC835
    bases = _walk_bases(lib, torch, n, T0, D)
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
    try:
        assert lib.zkhip_prepared_window_bits(h) == 20
        sc = cref.gen_scalars(62400, n, 0)
        dsc = torch.from_numpy(sc.view(np.int64)).cuda()
        out = torch.zeros(12, dtype=torch.int64, device="cuda")
        _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, dsc.data_ptr(), n, out.data_ptr(), None))
        torch.cuda.synchronize()
        assert np.array_equal(_aff(cref, out), _expect(cref, sc, T0, D))
    finally:
        _lib.check(lib.zkhip_release_bases(h))


@pytest.mark.parametrize("log_n,kind", [(17, 0), (18, 1), (20, 0)])
def test_msm_general_path_large(lib, cref, log_n, kind):
    """arbitrary (unregistered) device bases from 2^17 up: per-window bucket sets + the window fold"""
    import torch

    n = 1 << log_n
    T0, D = 0x5A4B534E41500002 + 17 * log_n, 0x9E3779B97F4A7C15F39CC0605CEDC835
    bases = _walk_bases(lib, torch, n, T0, D)
    sc = cref.gen_scalars(61700 + log_n, n, kind)
    dsc = torch.from_numpy(sc.view(np.int64)).cuda()
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_msm_g1_device(dsc.data_ptr(), bases.data_ptr(), n, out.data_ptr(), None))
    torch.cuda.synchronize()
    assert np.array_equal(_aff(cref, out), _expect(cref, sc, T0, D))


@pytest.mark.parametrize("n,off", [((1 << 21) + 5, 0), (1 << 21, 3), ((1 << 21) - 9, 0)])
def test_msm_shard_size_of_config4_prepared(lib, cref, n, off):
    """2^21 points per GPU (configs[4] = 2^24 over 8 GPUs): the wide sort's second kernel shape; ragged lengths and a sub-range"""
    import torch

    total = n + off + 3
    T0, D = 0x5A4B534E41500002 + 21, 0x9E3779B97F4A7C15F39CC0605CEDC835
    bases = _walk_bases(lib, torch, total, T0, D)
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), total, C.byref(h)))
    try:
        assert lib.zkhip_prepared_window_bits(h) == 20
        sc = cref.gen_scalars(62100 + off, n, 0)
        dsc = torch.from_numpy(sc.view(np.int64)).cuda()
        out = torch.zeros(12, dtype=torch.int64, device="cuda")
        _lib.check(lib.zkhip_msm_g1_prepared_device(h, off, dsc.data_ptr(), n, out.data_ptr(), None))
        torch.cuda.synchronize()
        assert np.array_equal(_aff(cref, out), _expect(cref, sc, (T0 + off * D) % O.R_MOD, D))
    finally:
        _lib.check(lib.zkhip_release_bases(h))


@pytest.mark.parametrize("n", [(1 << 19) - 1, (1 << 19) - 8, (1 << 19) + 1, (1 << 17) - 3])
@pytest.mark.parametrize("prepared", [False, True])
def test_msm_workspace_at_the_task_count_switch(lib, cref, n, prepared):
    """W n / 64 crosses 2^17 tasks just below n = 2^19 at c = 16: the workspace of a *fresh* context (grow-only scratch would hide
    an undersized estimate behind an earlier, larger call) must hold what the launcher lays out"""
    import torch

    lib.zkhip_shutdown()
    T0, D = 0x5A4B534E41500002 + 19, 0x9E3779B97F4A7C15F39CC0605CEDC835
    bases = _walk_bases(lib, torch, n, T0, D)
    sc = cref.gen_scalars(61900 + n % 97, n, 0)
    dsc = torch.from_numpy(sc.view(np.int64)).cuda()
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    if prepared:
        h = C.c_uint64(0)
        _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
        try:
            _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, dsc.data_ptr(), n, out.data_ptr(), None))
            torch.cuda.synchronize()
        finally:
            _lib.check(lib.zkhip_release_bases(h))
    else:
        _lib.check(lib.zkhip_msm_g1_device(dsc.data_ptr(), bases.data_ptr(), n, out.data_ptr(), None))
        torch.cuda.synchronize()
    assert np.array_equal(_aff(cref, out), _expect(cref, sc, T0, D))


@pytest.mark.parametrize("n,batch", [(8191, 129), (65535, 7), (32768 + 3, 33), (4093, 515)])
def test_msm_batched_non_power_of_two_fresh_context(lib, cref, n, batch):
    import torch

    lib.zkhip_shutdown()
    T0, D = 0x5A4B534E41500002 + 5, 0x9E3779B97F4A7C15F39CC0605CEDC835
    bases = _walk_bases(lib, torch, n, T0, D)
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
    try:
        stride = n + 5
        sc = cref.gen_scalars(61500 + batch, stride * batch, 0)
        dsc = torch.from_numpy(sc.view(np.int64)).cuda()
        out = torch.zeros(12 * batch, dtype=torch.int64, device="cuda")
        _lib.check(lib.zkhip_msm_g1_prepared_batch_device(h, 0, dsc.data_ptr(), n, batch, stride, out.data_ptr(), None))
        torch.cuda.synchronize()
        res = out.cpu().numpy().view(np.uint64).reshape(batch, 12)
        for b in sorted({0, 1, batch // 2, batch - 2, batch - 1}):
            assert np.array_equal(cref.jac_to_affine(res[b]), _expect(cref, sc[b * stride: b * stride + n], T0, D)), b
    finally:
        _lib.check(lib.zkhip_release_bases(h))


def test_msm_2p20_columns_of_equal_values_are_correct_and_not_serialised(lib, cref):
    """Columns real circuits commit to -- a selector (half ones), bits, an all-ones column, uniform values with 5 % equal to -1 -- at
    2^20 on the prepared path: each result against the structured-SRS identity, and none slower than 2.5 x the uniform case (their
    buckets hold up to 2^20 entries; with one thread writing such a bucket's task records an all-ones column took 8 x the uniform
    time: `profiles/r02_skewed_scalars.txt`)."""
    import time

    import torch

    n = 1 << 20
    T0, D = 0x5A4B534E41500999, 0x9E3779B97F4A7C15F39CC0605CEDC837
    t0m, dm = F.fr_encode([T0])[0], F.fr_encode([D])[0]
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
    one, minus_one = F.fr_encode([1])[0], F.fr_encode([O.R_MOD - 1])[0]
    uni = cref.gen_scalars(909, n, 0)
    rng = np.random.default_rng(4)
    cases = {"uniform": uni}
    sel = np.zeros((n, 4), dtype=np.uint64); sel[::2] = one
    cases["selector"] = sel
    bits = np.zeros((n, 4), dtype=np.uint64); bits[rng.random(n) < 0.5] = one
    cases["bits"] = bits
    cases["all ones"] = np.tile(one, (n, 1))
    mixed = uni.copy(); mixed[::20] = minus_one
    cases["5% equal to -1"] = mixed
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    times = {}
    try:
        for name, sc in cases.items():
            sc = np.ascontiguousarray(sc)
            dsc = torch.from_numpy(sc.view(np.int64)).cuda()
            run = lambda: _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, dsc.data_ptr(), n, out.data_ptr(), None))
            run()
            torch.cuda.synchronize()
            exp = cref.jac_to_affine(cref.scalar_mul(cref.expected_scalar(sc, T0, D), cref.generator()))
            assert np.array_equal(cref.jac_to_affine(out.cpu().numpy().view(np.uint64)), exp), name
            # device time (HIP events on the stream), best of three: a shared box may stretch any single run
            best = float("inf")
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    run()
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / 3)
            times[name] = best
    finally:
        _lib.check(lib.zkhip_release_bases(h))
    # the regression this guards against is not subtle: with one thread writing a heavy bucket's task records / one chain summing its partials,
    # the all-ones column took 8 x the uniform case (12.3 vs 1.5 ms).  A generous bound on best-of-three device times cannot flake on a noisy box.
    for name, t in times.items():
        assert t < 4.0 * times["uniform"], (name, times)


def test_batch_of_large_msms_overlapped_on_two_streams(lib, cref):
    """`zkhip_msm_g1_prepared_batch_device` with wide-window tables (n = 2^20): the vectors alternate between the caller's stream and the
    library's side stream; every result against the structured-SRS identity, also when the caller's stream is not the default one, with
    a padded stride and an odd batch; the host-buffer form (`zkhip_msm_g1_batch` on a registered array) takes the same path."""
    import torch

    n, batch, stride = 1 << 20, 3, (1 << 20) + 8
    T0, D = 0x5A4B534E41500AAA, 0x9E3779B97F4A7C15F39CC0605CEDC839
    t0m, dm = F.fr_encode([T0])[0], F.fr_encode([D])[0]
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
    assert lib.zkhip_prepared_window_bits(h) > 16
    vecs = [cref.gen_scalars(5100 + i, n, i % 2) for i in range(batch)]
    host = np.zeros((batch, stride, 4), dtype=np.uint64)
    for i, v in enumerate(vecs):
        host[i, :n] = v
    dsc = torch.from_numpy(host.view(np.int64)).cuda()
    exp = [cref.jac_to_affine(cref.scalar_mul(cref.expected_scalar(v, T0, D), cref.generator())) for v in vecs]
    try:
        for st in (None, torch.cuda.Stream()):
            out = torch.zeros((batch, 12), dtype=torch.int64, device="cuda")
            if st is not None:
                st.wait_stream(torch.cuda.current_stream())
            _lib.check(lib.zkhip_msm_g1_prepared_batch_device(h, 0, dsc.data_ptr(), n, batch, stride, out.data_ptr(), C.c_void_p(st.cuda_stream) if st else None))
            # results are complete when the CALLER's stream has drained (the side stream is joined to it)
            (st or torch.cuda.current_stream()).synchronize()
            got = out.cpu().numpy().view(np.uint64)
            for i in range(batch):
                assert np.array_equal(cref.jac_to_affine(got[i]), exp[i]), (i, st)
    finally:
        _lib.check(lib.zkhip_release_bases(h))
    # host-buffer form on a registered array
    hb = np.ascontiguousarray(bases.cpu().numpy().view(np.uint64).reshape(n, 8))
    _lib.check(lib.zkhip_register_bases(hb.ctypes.data, n))
    try:
        two = np.ascontiguousarray(np.stack(vecs[:2]))
        out = np.zeros((2, 12), dtype=np.uint64)
        _lib.check(lib.zkhip_msm_g1_batch(two.ctypes.data, hb.ctypes.data, n, 2, out.ctypes.data))
        for i in range(2):
            assert np.array_equal(cref.jac_to_affine(out[i]), exp[i])
    finally:
        _lib.check(lib.zkhip_unregister_bases(hb.ctypes.data))
