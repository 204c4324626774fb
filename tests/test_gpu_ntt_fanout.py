"""GPU (-m gpu): batched transforms spread over the devices of zkhip_init (second split: "different polynomials are
independent units ... batch parallelism, no collective"; workload: the 13 coset NTTs per wrapper proof,
/root/reference/aggregator/benches/wrapper_circuit.rs:21,61-68).

Every transform runs on ONE device with that device's own twiddle plan, so the results must be the same bytes whatever the device count.
The box has one card: the multi-device machinery (worker threads and per-device lanes for the host-buffer forms; fan streams, cross-device
events and peer copies for the `_device` forms) is rehearsed with three contexts on the same card (ZKHIP_TEST_DUPLICATE_DEVICES, the
pattern of tests/test_gpu_multi_shard.py) and compared with the single-device run and with the oracle's best_fft."""
import ctypes as C
import os

import numpy as np
import pytest

import zksnap_circuits_halo2_amd as Z
from zksnap_circuits_halo2_amd import _lib, fields as F

pytestmark = pytest.mark.gpu


class _Contexts:
    """zkhip_init over `ndev` contexts of card 0; restores the default single-device state on exit"""

    def __init__(self, lib, ndev, mode):
        self.lib, self.ndev, self.mode = lib, ndev, mode

    def __enter__(self):
        self.lib.zkhip_shutdown()
        if self.ndev > 1:
            os.environ["ZKHIP_TEST_DUPLICATE_DEVICES"] = "1"
        devs = (C.c_int * self.ndev)(*([0] * self.ndev))
        _lib.check(self.lib.zkhip_init(devs, self.ndev))
        assert self.lib.zkhip_device_count() == self.ndev
        _lib.check(self.lib.zkhip_set_ntt_fanout(self.mode))
        assert self.lib.zkhip_ntt_fanout() == self.mode
        return self

    def __exit__(self, *exc):
        self.lib.zkhip_shutdown()
        os.environ.pop("ZKHIP_TEST_DUPLICATE_DEVICES", None)
        _lib.check(self.lib.zkhip_init(None, 0))
        return False


def _host_forms(lib, polys, log_n, dom, coeffs):
    """the three host-buffer batch entry points on copies of the inputs; returns their outputs"""
    batch = len(polys)
    om = F.fr_encode([F.omega_for(log_n)])[0]
    omi = F.fr_encode([pow(F.omega_for(log_n), -1, F.R_MOD)])[0]
    div = F.fr_encode([pow(1 << log_n, -1, F.R_MOD)])[0]
    a = np.ascontiguousarray(np.stack(polys))
    _lib.check(lib.zkhip_ntt_fr_batch(a.ctypes.data, om.ctypes.data, log_n, batch))
    b = np.ascontiguousarray(np.stack(polys))
    _lib.check(lib.zkhip_ifft_scaled_batch(b.ctypes.data, omi.ctypes.data, log_n, div.ctypes.data, batch))
    c_in = np.ascontiguousarray(np.stack(coeffs))
    ext = np.zeros((len(coeffs), dom.extended_len(), 4), dtype=np.uint64)
    _lib.check(lib.zkhip_coeff_to_extended_batch(c_in.ctypes.data, dom.k, ext.ctypes.data, dom.extended_k, len(coeffs), dom.extended_omega.ctypes.data,
                                                 dom.g_coset.ctypes.data))
    return a, b, ext


@pytest.mark.parametrize("log_n,batch", [(14, 7), (17, 5), (11, 2), (14, 1)])
def test_host_batch_transforms_are_the_same_on_one_and_three_contexts(lib, cref, log_n, batch):
    """zkhip_ntt_fr_batch / zkhip_ifft_scaled_batch / zkhip_coeff_to_extended_batch: 1 context, 3 contexts (shares of 3 + 2 + 2, 2 + 2 + 1,
    1 + 1 + 0 polynomials and the batch of one), fan-out switched off on 3 contexts; against the oracle's best_fft and the single calls"""
    n = 1 << log_n
    polys = [cref.gen_scalars(9900 + 17 * log_n + b, n, b % 2) for b in range(batch)]
    k = log_n - 2
    dom = Z.EvaluationDomain(4, k)
    coeffs = [cref.gen_scalars(9950 + b, dom.n, (b + 1) % 2) for b in range(batch)]
    with _Contexts(lib, 1, 1):
        one = _host_forms(lib, polys, log_n, dom, coeffs)
        singles_ext = [dom.coeff_to_extended(c) for c in coeffs]
    om = F.fr_encode([F.omega_for(log_n)])[0]
    for b, p in enumerate(polys):
        ref = p.copy()
        cref.best_fft(ref, om, log_n, 4)
        assert np.array_equal(one[0][b], ref), b
        assert np.array_equal(one[2][b], singles_ext[b]), b
    with _Contexts(lib, 3, 1):
        three = _host_forms(lib, polys, log_n, dom, coeffs)
    with _Contexts(lib, 3, 0):
        off = _host_forms(lib, polys, log_n, dom, coeffs)
    for x, y, z in zip(one, three, off):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    # iNTT(NTT(p)) = p through the two batch calls on three contexts
    with _Contexts(lib, 3, 1):
        a = one[0].copy()
        omi = F.fr_encode([pow(F.omega_for(log_n), -1, F.R_MOD)])[0]
        div = F.fr_encode([pow(1 << log_n, -1, F.R_MOD)])[0]
        _lib.check(lib.zkhip_ifft_scaled_batch(a.ctypes.data, omi.ctypes.data, log_n, div.ctypes.data, batch))
    assert np.array_equal(a, np.stack(polys))


def _device_forms(lib, polys, log_n, dom, coeffs):
    """the `_device` batch entry points with strides wider than the polynomials; returns host copies of the outputs"""
    import torch

    batch, n = len(polys), 1 << log_n
    stride = n + 24
    om = F.fr_encode([F.omega_for(log_n)])[0]
    omi = F.fr_encode([pow(F.omega_for(log_n), -1, F.R_MOD)])[0]
    div = F.fr_encode([pow(1 << log_n, -1, F.R_MOD)])[0]
    s = torch.cuda.current_stream().cuda_stream

    def strided(vs, st):
        d = torch.zeros(len(vs) * st * 4, dtype=torch.int64, device="cuda")
        for b, p in enumerate(vs):
            d[b * st * 4:(b * st + p.shape[0]) * 4] = torch.from_numpy(p.view(np.int64).reshape(-1)).cuda()
        return d

    d_a = strided(polys, stride)
    _lib.check(lib.zkhip_ntt_fr_batch_device(d_a.data_ptr(), om.ctypes.data, log_n, batch, stride, s))
    d_b = strided(polys, stride)
    _lib.check(lib.zkhip_ifft_scaled_batch_device(d_b.data_ptr(), omi.ctypes.data, log_n, div.ctypes.data, batch, stride, s))
    en, a_stride, e_stride, o_stride = dom.extended_len(), dom.n + 8, dom.extended_len() + 16, 3 * dom.n + 4
    d_c = strided(coeffs, a_stride)
    d_e = torch.zeros(batch * e_stride * 4, dtype=torch.int64, device="cuda")
    d_o = torch.zeros(batch * o_stride * 4, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_coeff_to_extended_device(d_c.data_ptr(), a_stride, dom.k, d_e.data_ptr(), e_stride, dom.extended_k, batch,
                                                  dom.extended_omega.ctypes.data, dom.g_coset.ctypes.data, s))
    _lib.check(lib.zkhip_extended_to_coeff_device(d_e.data_ptr(), e_stride, dom.extended_k, dom.extended_omega_inv.ctypes.data,
                                                  dom.extended_ifft_divisor.ctypes.data, dom.g_coset.ctypes.data, d_o.data_ptr(), o_stride, 3 * dom.n, batch, s))
    torch.cuda.synchronize()
    host = lambda d, st: d.cpu().numpy().view(np.uint64).reshape(batch, st, 4)
    return host(d_a, stride), host(d_b, stride), host(d_e, e_stride), host(d_o, o_stride)


@pytest.mark.parametrize("log_n,batch", [(14, 7), (17, 4), (13, 2)])
def test_device_batch_transforms_with_peer_copies_on_three_contexts(lib, cref, log_n, batch):
    """fan-out mode 2: the secondary contexts pull their polynomials with peer copies, transform them on their fan streams and push the
    results back behind events; same bytes as the primary-only run, padding between the polynomials untouched, coset round trip exact"""
    n = 1 << log_n
    polys = [cref.gen_scalars(9800 + 13 * log_n + b, n, b % 2) for b in range(batch)]
    dom = Z.EvaluationDomain(4, log_n - 2)
    coeffs = [cref.gen_scalars(9850 + b, dom.n, (b + 1) % 2) for b in range(batch)]
    with _Contexts(lib, 1, 1):
        one = _device_forms(lib, polys, log_n, dom, coeffs)
    with _Contexts(lib, 3, 2):
        three = _device_forms(lib, polys, log_n, dom, coeffs)
        again = _device_forms(lib, polys, log_n, dom, coeffs)        # the plans, fan streams and scratch of the first call reused
    with _Contexts(lib, 3, 1):
        host_only_mode = _device_forms(lib, polys, log_n, dom, coeffs)   # mode 1 leaves the `_device` forms on the primary
    for x, y, z, w in zip(one, three, again, host_only_mode):
        assert np.array_equal(x, y) and np.array_equal(x, z) and np.array_equal(x, w)
    om = F.fr_encode([F.omega_for(log_n)])[0]
    for b, p in enumerate(polys):
        ref = p.copy()
        cref.best_fft(ref, om, log_n, 4)
        assert np.array_equal(three[0][b, :n], ref), b
        assert not three[0][b, n:].any()
        assert np.array_equal(three[3][b, :dom.n], coeffs[b]) and not three[3][b, dom.n:].any(), b


def test_fanout_mode_is_validated_and_reported(lib):
    assert lib.zkhip_set_ntt_fanout(3) != 0 and lib.zkhip_set_ntt_fanout(-1) != 0
    assert b"set_ntt_fanout" in lib.zkhip_last_error()
    _lib.check(lib.zkhip_set_ntt_fanout(1))
    assert lib.zkhip_ntt_fanout() == 1
