"""CPU, world_size 2, gloo: the sharded MSM's partitioning + all_gather + fold, with the oracle standing in for the
per-rank GPU MSM (the data path itself is covered by the -m gpu tests)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from oracle import cpu_ref as Cr
    from zksnap_circuits_halo2_amd.multi_gpu import shard_range, sharded_msm

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    bases, t0, d = Cr.gen_bases(321, n)
    sc = Cr.gen_scalars(654, n, 1)
    lo, hi = shard_range(n, rank, world)

    def fold(parts):
        acc = parts[0].copy()
        for p in parts[1:]:
            acc = Cr.jac_add(acc, np.ascontiguousarray(p))
        return acc

    out = sharded_msm(sc[lo:hi], bases[lo:hi], local_msm=lambda s, b: Cr.best_multiexp(np.ascontiguousarray(s), np.ascontiguousarray(b), 1), fold=fold)
    exp = Cr.jac_to_affine(Cr.scalar_mul(Cr.expected_scalar(sc, t0, d), Cr.generator()))
    q.put((rank, bool(np.array_equal(Cr.jac_to_affine(out), exp))))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_msm_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, n = 2, 777   # ragged: 389 + 388
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
