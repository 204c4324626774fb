#!/usr/bin/env python3
"""bench.py -- BN254 G1 MSM throughput on MI355X (BASELINE.json metric, configs[1]: 2^20 random points/scalars).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one MSM over the rank's shard (2^20 points per GPU, weak scaling: the N-GPU job is one 2^20*N-point MSM
sharded by point range) + the partial-sum exchange (all_gather of 96-byte Jacobian partials over RCCL, folded on
device).  Scalars and bases are resident in HBM before the timed region; prints ONE JSON line on rank 0.
Extra keys: "roofline" (dominant kernel = bucket accumulation, HIP-event timed inside the library on the launch
stream), "cpu_baseline" (oracle/cpu_ref.c = restatement of the reference's best_multiexp on the host cores),
"phases_ms", "ntt", "wrapper_replay" (MSM+NTT call mix of one wrapper-circuit proof at k = 22, device-resident), "value_2p22",
"wrapper_replay_k24" (the same mix at k = 24 = BASELINE configs[4], one card + the 8-card projection), "prover_flow_k13_wide" / "prover_flow_k15_wide".
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

R_TOP = 0x30644E72E131A029  # top limb of r: limbs with top < R_TOP are canonical Fr values


def synth_scalars(n: int, seed: int, kind: str = "uniform") -> np.ndarray:
    """(n,4) uint64 canonical limbs = uniformly random Fr elements in the Montgomery memory format (a uniformly random
    canonical limb pattern is the Montgomery form of a uniformly random field element).  No oracle involved."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    a[:, 3] = rng.integers(0, R_TOP, size=n, dtype=np.uint64)
    if kind == "witness":  # 60 % zero, 30 % small (< 2^88 as *integers*: needs Montgomery conversion, done on host bigints for a sample only)
        sel = rng.integers(0, 10, size=n)
        a[sel < 6] = 0
    return a


def profile_read(lib, counts: dict = None):
    """phases of the last profiled call: [(name, ms)] with the marks of one name summed (the wide MSM marks accumulate / combine once per
    bucket-range piece); `counts` receives how many marks each name had"""
    ms = (C.c_double * 32)()
    names = ((C.c_char * 64) * 32)()
    k = lib.zkhip_profile_read(ms, names, 32)
    per = {}
    for i in range(max(k, 0)):
        nm = names[i].value.decode()
        per[nm] = per.get(nm, 0.0) + ms[i]
        if counts is not None:
            counts[nm] = counts.get(nm, 0) + 1
    return list(per.items())


# The card drops its clocks when it idles (host-side setup between measurements is enough) and takes 30-40 ms of continuous work to reach them
# again: back-to-back 2^20 MSMs after an idle second take 1.59-1.62 ms for calls 0-4, 1.50 for 5-9, 1.43 for 10-19 and 1.374 from call 40 on; the
# 2^24 NTT 2.22 / 1.99 / 1.85 / 1.81 ms (tools/ramp_probe.py, profiles/r03_clock_ramp.txt).  Throughput is a steady-state figure, so every timed
# region is preceded by untimed calls of the same operation until PREWARM_MS of wall time have passed; the line states it ("clock_prewarm").
PREWARM_MS = 100.0


def prewarm(fn, torch, min_calls: int = 1, ms: float = PREWARM_MS, max_calls: int = 400) -> int:
    """untimed calls of fn until `ms` of wall time (and at least min_calls) have gone by; returns the number of calls"""
    t = time.perf_counter()
    calls = 0
    while calls < max_calls and (calls < min_calls or (time.perf_counter() - t) * 1e3 < ms):
        fn()
        calls += 1
        if calls % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return calls


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log-n", type=int, default=20, help="log2 points per GPU (BASELINE configs[1] = 20)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the NTT / wrapper-replay extras")
    ap.add_argument("--no-general-path", action="store_true", help="skip the unregistered-bases MSM (rocprofv3 runs: keeps per-kernel averages to the headline path)")
    ap.add_argument("--c-abi-config4", type=int, default=0, metavar="NDEV",
                    help="internal: run ONLY the single-process C-ABI leg of configs[4] over devices 0 .. NDEV-1 and print its dict as one JSON line (the N > 1 "
                         "run starts this in a child process: the leg drives several cards from one process -- peer access, peer copies -- which has never "
                         "run on real multi-GPU hardware, and a fault there must not take the bench line with it)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="CPU rehearsal of the --gpus N launcher: the ranks rendezvous over gloo, time an empty step and rank 0 prints a line that is "
                         "marked as a self-test (no GPU work, no metric)")
    return ap.parse_args(argv)


def structured_point(lib, _lib, F, torch, d_scalars, n, t0: int, d: int, stream, dev):
    """[sum_i a_i (t0 + i d)] G as affine Montgomery words (8 x u64) -- what an MSM of the scalars a against the walk bases (t0 + i d) G must be.
    Computed WITHOUT the MSM kernels and without the oracle: sum a_i = p(1) (zkhip_fr_eval_polynomial_device), sum i a_i = p'(1) = q(1) for
    q = (p - p(1)) / (X - 1) (zkhip_fr_kate_division_device + one more evaluation), the two 256-bit results combined on the host, and one
    fixed-base multiplication of the generator (zkhip_g1_fixed_base_mul_device)."""
    one = F.fr_encode([1])[0]
    d_q = torch.empty(max(n - 1, 1) * 4, dtype=torch.int64, device=dev)
    d_ev = torch.zeros(8, dtype=torch.int64, device=dev)
    _lib.check(lib.zkhip_fr_eval_polynomial_device(d_scalars.data_ptr(), n, one.ctypes.data, d_ev.data_ptr(), stream))
    if n > 1:
        _lib.check(lib.zkhip_fr_kate_division_device(d_scalars.data_ptr(), n, one.ctypes.data, d_q.data_ptr(), stream))
        _lib.check(lib.zkhip_fr_eval_polynomial_device(d_q.data_ptr(), n - 1, one.ctypes.data, d_ev.data_ptr() + 32, stream))
    _lib.check(lib.zkhip_stream_sync(stream))
    p1, dp1 = F.fr_decode(d_ev.cpu().numpy().view(np.uint64).reshape(2, 4))
    k = (t0 * p1 + d * dp1) % F.R_MOD
    d_k = torch.from_numpy(F.fr_encode([k]).view(np.int64)).to(dev)
    d_pt = torch.zeros(8, dtype=torch.int64, device=dev)
    _lib.check(lib.zkhip_g1_fixed_base_mul_device(d_k.data_ptr(), 1, d_pt.data_ptr(), stream))
    _lib.check(lib.zkhip_stream_sync(stream))
    return d_pt.cpu().numpy().view(np.uint64).copy()


def affine_words(lib, _lib, jac_words: np.ndarray) -> np.ndarray:
    """(m, 12) Jacobian Montgomery words -> (m, 8) affine (zkhip_g1_batch_normalize: canonical words, identity = zeros)"""
    jac = np.ascontiguousarray(jac_words, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros((jac.shape[0], 8), dtype=np.uint64)
    _lib.check(lib.zkhip_g1_batch_normalize(jac.ctypes.data, jac.shape[0], out.ctypes.data))
    return out


def verify_exchange(lib, _lib, F, torch, dist, dev, stream, cpu_group, world, d_scalars, n, t0: int, d: int, d_partial, d_gather, d_final,
                    check_devices: bool = True) -> dict:
    """Evidence that an N-rank line is what it says, computed after the timed region from the buffers of its last step:
      ranks_seen      device-side all_reduce(SUM) of ones over the data-path backend (RCCL at N > 1)
      devices         every rank's device (name, PCI bus id, uuid), gathered on the host group; two ranks on one device are refused unless
                      ZKHIP_BENCH_ALLOW_SHARED_DEVICE=1 (rehearsals on a one-card box)
      result_checked  every rank: its partial sum == the structured identity of ITS walk and scalars (structured_point: no MSM kernel involved);
                      rank 0: the gathered slots hold exactly the ranks' partials (compared with what each rank reports over the host group) and
                      the fold equals zkhip_g1_sum of the gathered partials recomputed on the host path"""
    rank = dist.get_rank()
    ones = torch.ones(1, dtype=torch.int64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(ones)
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "device_index": dev.index, "name": props.name,
          "pci": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0)),
          "uuid": str(getattr(props, "uuid", ""))}
    want = structured_point(lib, _lib, F, torch, d_scalars, n, t0, d, stream, dev)
    part = d_partial.cpu().numpy().view(np.uint64)[:12].copy()
    me["partial_ok"] = bool(np.array_equal(affine_words(lib, _lib, part)[0], want))
    me["partial"] = part.tolist()
    everyone = [None] * world
    dist.all_gather_object(everyone, me, group=cpu_group)
    out = {"ranks_seen": int(ones.item()), "backend": dist.get_backend(),
           "devices": [{k: e[k] for k in ("rank", "device_index", "name", "pci", "uuid")} for e in everyone]}
    ids = [(e["pci"], e["uuid"], e["device_index"]) for e in everyone]
    out["distinct_devices"] = len(set(ids)) == world
    if check_devices and not out["distinct_devices"] and os.environ.get("ZKHIP_BENCH_ALLOW_SHARED_DEVICE") != "1":
        fail(f"{world} ranks but only {len(set(ids))} distinct device(s) {sorted(set(ids))}: refusing to report an n_gpus = {world} line", 4)
    partial_ok = all(e["partial_ok"] for e in everyone)
    if rank == 0:
        width = d_partial.numel()
        gathered = d_gather.cpu().numpy().view(np.uint64).reshape(world, width)[:, :12]
        slots_ok = all(np.array_equal(gathered[r], np.array(everyone[r]["partial"], dtype=np.uint64)) for r in range(world))
        refold = np.zeros(12, dtype=np.uint64)
        g = np.ascontiguousarray(gathered)
        _lib.check(lib.zkhip_g1_sum(g.ctypes.data, world, refold.ctypes.data))
        fold_ok = bool(np.array_equal(affine_words(lib, _lib, refold)[0], affine_words(lib, _lib, d_final.cpu().numpy().view(np.uint64)[:12])[0]))
        out["result_checked"] = {"every_rank_partial_equals_its_structured_identity": bool(partial_ok), "gathered_slots_are_the_ranks_partials": bool(slots_ok),
                                 "fold_equals_sum_of_gathered_partials": fold_ok, "ok": bool(partial_ok and slots_ok and fold_ok)}
        if not out["result_checked"]["ok"]:
            fail(f"N = {world} result check failed: {out['result_checked']}", 5)
    return out


def fail(msg: str, code: int = 2):
    print("bench.py: " + msg, file=sys.stderr, flush=True)
    sys.exit(code)


def visible_gpus_without_hip():
    """AMD GPUs this process could open, counted WITHOUT initialising the HIP / HSA runtime (the launcher parent must stay clean of it: on ROCm
    wheels `torch.cuda.device_count()` may fall through to hipGetDeviceCount and open /dev/kfd).  The KFD topology in sysfs lists one node per
    agent of the HOST (GPUs: simd_count > 0) whatever the container may use, so a GPU counts only if its DRM render node can actually be opened
    (a device cgroup refuses the open of the cards that are not this container's); *_VISIBLE_DEVICES lists cap the count.  None when the
    topology cannot be read (the per-rank check in main() still refuses a rank without its device)."""
    root = os.environ.get("ZKHIP_BENCH_KFD_NODES", "/sys/class/kfd/kfd/topology/nodes")      # (the overrides exist for tests/test_bench_launcher.py)
    dri = os.environ.get("ZKHIP_BENCH_DRI_DIR", "/dev/dri")
    if not os.path.isdir(root):
        return 0                                      # no KFD driver: no AMD GPU
    count = 0
    try:
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) <= 0:
                continue
            minor = int(props.get("drm_render_minor", "-1"))
            if minor < 0:
                continue
            try:
                fd = os.open(os.path.join(dri, f"renderD{minor}"), os.O_RDWR)      # opening the render node starts nothing: no KFD process, no queues
                os.close(fd)
                count += 1
            except OSError:
                pass
    except (OSError, ValueError):
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            count = min(count, len([x for x in v.split(",") if x.strip() != ""]))
    return count


def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` outside a torchrun environment: start the N ranks ourselves (one process per GPU) and return the job's exit
    status.  This process never touches HIP (the device count comes from sysfs, torch is not even imported), it starts the ranks as child
    processes -- never a re-exec -- and rank 0's JSON line reaches stdout through the inherited descriptor."""
    import socket
    import subprocess

    if not args.launcher_selftest:
        have = visible_gpus_without_hip()
        if have is not None and have < args.gpus:
            fail(f"--gpus {args.gpus} asked for, {have} HIP device(s) visible: refusing to run a {args.gpus}-GPU measurement on fewer GPUs", 3)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # RCCL / dmabuf IPC on this driver
    env["ZKHIP_BENCH_LAUNCHED_BY"] = "bench.py"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def launcher_selftest(rank: int, world: int, args) -> None:
    """What a rank does under --launcher-selftest: the rendezvous, the barrier-bracketed timing and the max over ranks of the real step
    loop, on gloo, around an empty step."""
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    t = time.perf_counter()
    for _ in range(args.steps):
        pass
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t], dtype=torch.float64)
    ranks = torch.tensor([1], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(ranks)
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "ranks_seen": int(ranks.item()), "steps": args.steps, "warmup": args.warmup,
                          "launched_by": os.environ.get("ZKHIP_BENCH_LAUNCHED_BY", "external torchrun"), "metric": None, "value": None}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main() -> None:
    args = parse_args()
    # ---- who starts the ranks ---------------------------------------------------------------------------------------------------------
    # The metric is "Mpoints/sec at 1/2/4/8 MI355X": --gpus N must mean N ranks on N GPUs or no number at all.  Under torchrun (the driver's
    # form for N > 1) WORLD_SIZE is set and must equal --gpus; without it, N > 1 starts the ranks here, before anything touches the GPU.
    if args.c_abi_config4 > 0:
        import torch

        from zksnap_circuits_halo2_amd import _lib, fields as F

        torch.cuda.set_device(0)
        dev0 = torch.device("cuda", 0)
        try:
            out = c_abi_config4(_lib.load(), _lib, F, torch, dev0, torch.cuda.current_stream().cuda_stream, devices=list(range(args.c_abi_config4)), shards=args.c_abi_config4)
        except Exception as exc:
            out = {"error": repr(exc)}
        print(json.dumps(out), flush=True)
        return
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        fail(f"--gpus {args.gpus}: need at least one GPU")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        fail(f"--gpus {args.gpus} but WORLD_SIZE={env_world}: the launcher's rank count and --gpus disagree; refusing to report n_gpus for a job of another size")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if args.launcher_selftest:
        launcher_selftest(rank, world, args)
        return

    import torch
    import torch.distributed as dist

    if torch.cuda.device_count() <= local_rank or torch.cuda.device_count() < world:
        fail(f"rank {rank}: local rank {local_rank} of {world} needs {world} visible HIP devices, {torch.cuda.device_count()} found: refusing to run a "
             f"{world}-GPU measurement on fewer GPUs", 3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cpu_group = None
    # ZKHIP_BENCH_FORCE_DIST=1: the rendezvous, barriers, all_gather and all_reduce of the N > 1 path run on RCCL even with ONE rank (a rehearsal of
    # the multi-GPU control flow on a one-GPU box: tests/test_gpu_multi_shard.py); the line says so ("rccl_rehearsal")
    use_dist = world > 1 or os.environ.get("ZKHIP_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group("nccl", device_id=dev)
        cpu_group = dist.new_group(backend="gloo")      # host-side waits that must not occupy the GPUs (the single-process leg at the end)

    from zksnap_circuits_halo2_amd import _lib, fields as F
    from zksnap_circuits_halo2_amd.multi_gpu import gather_fold_device

    lib = _lib.load()
    devs = (C.c_int * 1)(local_rank)
    _lib.check(lib.zkhip_init(devs, 1))
    stream = torch.cuda.current_stream().cuda_stream

    n = 1 << args.log_n
    # ---- inputs, resident in HBM --------------------------------------------------------------------------
    t0 = F.fr_encode([0x5A4B534E41500002 + 7919 * rank])[0]
    dd = F.fr_encode([0x9E3779B97F4A7C15F39CC0605CEDC835])[0]
    d_bases = torch.empty(n * 8, dtype=torch.int64, device=dev)
    _lib.check(lib.zkhip_g1_gen_walk_device(t0.ctypes.data, dd.ctypes.data, n, d_bases.data_ptr(), stream))
    # the drop-in registers `ParamsKZG::{g, g_lagrange}` once (zkhip_register_bases); the device-side equivalent:
    handle = C.c_uint64(0)
    torch.cuda.synchronize()
    t_prep = time.perf_counter()
    _lib.check(lib.zkhip_prepare_bases_device(d_bases.data_ptr(), n, C.byref(handle)))
    torch.cuda.synchronize()
    t_prep = time.perf_counter() - t_prep
    d_scalars = torch.from_numpy(synth_scalars(n, 0x5A4B534E41500003 + rank).view(np.int64)).to(dev)
    d_out = torch.zeros(12, dtype=torch.int64, device=dev)          # 96-byte Jacobian result
    # Exchange per step: one 96-byte all_gather + a fold of the N partials, enqueued behind the MSM on the same stream.  (Running it on a
    # side stream under the next step's MSM was measured at N = 1 with the gather degenerated to a copy -- ZKHIP_BENCH_FORCE_EXCHANGE=1
    # -- and is slower: 599 vs 612 Mpoints/s, against 616 without any exchange; the cross-stream events cost more than the fold.)
    exchange = use_dist or os.environ.get("ZKHIP_BENCH_FORCE_EXCHANGE") == "1"
    d_gather = torch.zeros(12 * world, dtype=torch.int64, device=dev)
    d_final = torch.zeros(12, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    def step():
        _lib.check(lib.zkhip_msm_g1_prepared_device(handle, 0, d_scalars.data_ptr(), n, d_out.data_ptr(), stream))
        if exchange:   # every rank gathers the partials and folds them
            gather_fold_device(d_out, d_gather, d_final, stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # the W warm-up steps of the contract, then untimed steps until the card has been busy for PREWARM_MS (see prewarm): W = 5 steps are 7 ms
    # (the MSM alone, not step(): the count is time-based and differs from rank to rank, and step() holds a collective)
    def msm_only():
        _lib.check(lib.zkhip_msm_g1_prepared_device(handle, 0, d_scalars.data_ptr(), n, d_out.data_ptr(), stream))

    # (round 4, advisor) first the figure WITHOUT the pre-warm, for comparison with the lines of rounds 1-2: K steps straight after the W warm-up
    # steps, bracketed the same way -- reported as value_cold, never as value
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t_cold = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed_cold = time.perf_counter() - t_cold
    # the rendezvous comes BEFORE the pre-warm so that no host-side idle gap separates the pre-warm from the timed region (each rank pre-warms
    # for the same wall time, then starts its K steps; the closing barrier and the max over ranks absorb the skew)
    if use_dist:
        dist.barrier()
    extra_warm = prewarm(msm_only, torch, 0) if os.environ.get("ZKHIP_BENCH_NO_PREWARM") != "1" else 0
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    if use_dist:
        t = torch.tensor([elapsed, elapsed_cold], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed_cold = float(t[0].item()), float(t[1].item())

    ms_per_step = elapsed / args.steps * 1e3
    mpoints = world * n * args.steps / elapsed / 1e6
    if exchange and world == 1:   # forced exchange at N = 1: the fold of one partial must be that partial (as an affine point)
        assert F.g1_decode_jacobian(d_final.cpu().numpy().view(np.uint64)[:12]) == F.g1_decode_jacobian(d_out.cpu().numpy().view(np.uint64)[:12])

    result = {
        "metric": "BN254 G1 MSM Mpoints/sec",
        "value": round(mpoints, 3),
        "unit": "Mpoints/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "value_cold": round(world * n * args.steps / elapsed_cold / 1e6, 3),
        "clock_prewarm": {"untimed_steps_after_the_warmup": extra_warm + args.steps, "ms": PREWARM_MS, "value_cold_is": "the K steps straight after the W warm-up steps (no pre-warm), same bracketing",
                          "why": "the card needs 30-40 ms of continuous work to reach its clocks after idling; steady-state throughput is the metric (DESIGN.md section 7)"},
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32x9 (254-bit modular integer, radix 2^29)",
        "data": "synthetic",
        "config": {"workload": f"BN254 G1 MSM, 2^{args.log_n} random points/scalars per GPU (BASELINE configs[1]), "
                               f"point-range sharded x{world}, inputs resident in HBM, bases registered once "
                               "(prepared fixed-base table, as under ParamsKZG::commit)",
                   "points_per_gpu": n, "prepare_bases_ms_one_time": round(t_prep * 1e3, 2),
                   "parallelism": f"point-range shard x{world} + all_gather(96 B) + fold"},
    }

    if use_dist:
        # the N-rank line carries its own evidence (ranks, devices, results): see verify_exchange
        ver = verify_exchange(lib, _lib, F, torch, dist, dev, stream, cpu_group, world, d_scalars, n, 0x5A4B534E41500002 + 7919 * rank,
                              0x9E3779B97F4A7C15F39CC0605CEDC835, d_out, d_gather, d_final)
        result.update({k: ver[k] for k in ("ranks_seen", "devices", "distinct_devices", "result_checked") if k in ver})
        result["exchange_backend"] = ver["backend"]
    elif world == 1:
        want = structured_point(lib, _lib, F, torch, d_scalars, n, 0x5A4B534E41500002, 0x9E3779B97F4A7C15F39CC0605CEDC835, stream, dev)
        ok = bool(np.array_equal(affine_words(lib, _lib, d_out.cpu().numpy().view(np.uint64)[:12])[0], want))
        result["result_checked"] = {"msm_equals_structured_identity": ok, "ok": ok}
        if not ok:
            fail("the timed MSM's result is not [sum a_i (t0 + i d)] G", 5)

    if rank == 0:
        # ---- roofline of the dominant kernel (bucket accumulation), HIP events on the launch stream ---------
        # Measured in the headline's state (round 4 review, item 4): prepared MSMs only, back to back behind the timed region's own pre-warm,
        # the library's phase events APPENDED call after call (zkhip_profile_enable(2)) and read back once at the end -- no host read-back and
        # no other work between the calls.  The general path's phases come from their own loop below.
        def phase_loop(call, reps):
            prewarm(call, torch, 0, ms=30.0)
            lib.zkhip_profile_enable(2)
            for _ in range(reps):
                call()
            cap = 32 * reps
            ms = (C.c_double * cap)()
            names = ((C.c_char * 64) * cap)()
            call_of = (C.c_int * cap)()
            k = lib.zkhip_profile_read_calls(ms, names, call_of, cap)
            lib.zkhip_profile_enable(0)
            per_call = {}
            for i in range(max(k, 0)):
                d = per_call.setdefault(call_of[i], {})
                nm = names[i].value.decode()
                d[nm] = d.get(nm, 0.0) + ms[i]
            series = {}
            for d in per_call.values():
                for nm, v in d.items():
                    series.setdefault(nm, []).append(v)
            # per phase: the MEDIAN over the calls (one call in a dozen shows one launch of one phase three times its usual length: a clock or
            # scheduling event on the box, not the kernel; dropping only the maximum would bias the figure downward)
            return {nm: float(np.median(v)) for nm, v in series.items()}, len(per_call)

        acc, acc_calls = phase_loop(msm_only, 16)
        gen = {}
        if not args.no_general_path:
            gen, _ = phase_loop(lambda: _lib.check(lib.zkhip_msm_g1_device(d_scalars.data_ptr(), d_bases.data_ptr(), n, d_out.data_ptr(), stream)), 8)
        # arbitrary (unregistered) bases: per-window bucket sets + window fold
        if gen:
            # the equal-work replacement of best_multiexp (arbitrary, unregistered bases: per-window bucket sets + window fold), next to the
            # headline, which is the ParamsKZG::commit case (bases fixed per params object -> prepared table built once, outside the timed region)
            result["value_general_path"] = round(n / sum(gen.values()) / 1e3, 2)
            result["general_path"] = {"ms": round(sum(gen.values()), 4), "Mpoints_per_s": round(n / sum(gen.values()) / 1e3, 2),
                                      "phases_ms": {k: round(v, 4) for k, v in gen.items()},
                                      "note": "in-library HIP events, 8 calls back to back in their own loop (medians), bases device-resident but not prepared"}
        _lib.check(lib.zkhip_msm_g1_prepared_device(handle, 0, d_scalars.data_ptr(), n, d_out.data_ptr(), stream))
        torch.cuda.synchronize()
        # throughput of INDEPENDENT MSMs (a prover commits several columns per round) through the batch entry point: for tables with wide
        # windows the library alternates the vectors between the caller's stream and its own side stream, so that one MSM's latency-bound
        # sort and reduction tail run under the other's accumulation.  Beside the headline, which stays one MSM after the other.
        try:
            if args.no_extras:       # (the profiled command: its kernel statistics should hold single-stream launches only)
                raise RuntimeError("skipped (--no-extras)")
            kb = 8
            many = d_scalars.reshape(-1, 4).repeat(kb, 1).contiguous()
            outs_b = torch.zeros(kb * 12, dtype=torch.int64, device=dev)
            run_b = lambda: _lib.check(lib.zkhip_msm_g1_prepared_batch_device(handle, 0, many.data_ptr(), n, kb, n, outs_b.data_ptr(), stream))
            run_b()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            reps_b = max(1, args.steps // kb)
            for _ in range(reps_b):
                run_b()
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t2) / (reps_b * kb)
            same = all(F.g1_decode_jacobian(outs_b.cpu().numpy().view(np.uint64)[12 * i:12 * i + 12]) == F.g1_decode_jacobian(d_out.cpu().numpy().view(np.uint64)[:12]) for i in range(kb))
            result["value_batch_of_8"] = round(n / dt2 / 1e6, 2)
            result["batch_of_8"] = {"ms_per_msm": round(dt2 * 1e3, 4), "Mpoints_per_s": result["value_batch_of_8"], "results_equal_the_single_msm": bool(same),
                                    "note": "8 independent 2^%d MSMs per call of zkhip_msm_g1_prepared_batch_device (pairs overlap on two streams inside the library); not the headline" % args.log_n}
            del many
        except Exception as exc:   # an extra: never fail the bench line
            result["batch_of_8"] = {"error": repr(exc)}
        t_acc = acc.get("accumulate", float("nan"))
        alg_bytes = 96.0 * n                                       # 64 B affine base + 32 B scalar per point
        achieved = alg_bytes / (t_acc * 1e-3) / 1e9
        traffic, traffic_source, whole_msm_traffic = None, None, None
        try:   # PMC-measured HBM bytes of this kernel at this size: NOT measured by this run -- collected in separate rocprofv3 --pmc passes of
               # this same command (tools/collect_profiles.sh) and committed; the file names the commit it was collected at
            if args.log_n == 20:
                for cand in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
                    path = os.path.join(ROOT, "profiles", cand)
                    if os.path.exists(path):
                        rec = json.load(open(path))
                        traffic = rec["k_accumulate"]["hbm_bytes_per_launch"]
                        whole = rec.get("whole_msm")      # round 4: every kernel of the MSM, launches per MSM x bytes per launch, summed
                        if whole:
                            whole_msm_traffic = {"hbm_bytes_raw_fetch": whole["hbm_bytes_raw_fetch"], "hbm_bytes_x2_fetch": whole["hbm_bytes_x2_fetch"],
                                                 "over_algorithmic_raw": round(whole["hbm_bytes_raw_fetch"] / (96.0 * n), 2),
                                                 "per_kernel_x2": {k.replace("zkhip::", ""): v["hbm_bytes_per_msm_x2_fetch"] for k, v in rec.get("kernels_per_msm", {}).items()
                                                                   if v["hbm_bytes_per_msm_x2_fetch"] >= 1 << 20},
                                                 "source": f"profiles/{cand} (same passes as `traffic`)"}
                        traffic_source = f"profiles/{cand}@{rec.get('commit', 'round-1 head')} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE per launch -- the x2 is the guide's gfx950 correction for coalesced streams, an assumption for 64-byte gathers: the file also holds the raw counter; a committed constant, not a live counter)"
                        break
        except Exception:
            traffic, traffic_source = None, None
        # the same kernel's average duration in the committed rocprofv3 --kernel-trace --stats summary of this command (tools/collect_profiles.sh)
        frac_rocprof, rocprof_src, frac_rocprof_median = None, None, None
        try:
            import csv as _csv
            for cand in ("r05_rocprofv3_kernel_stats_bench_msm2p20.csv", "r04_rocprofv3_kernel_stats_bench_msm2p20.csv"):
                path = os.path.join(ROOT, "profiles", cand)
                if args.log_n == 20 and os.path.exists(path):
                    for row in _csv.DictReader(open(path)):
                        if "k_accumulate<true>" in row["Name"]:
                            frac_rocprof = round(alg_bytes / (float(row["AverageNs"]) * 1e-9) / 1e9 / 8000.0, 5)
                            rocprof_src = f"profiles/{cand}: average of {row['Calls']} launches = {float(row['AverageNs']) / 1e6:.4f} ms"
                            if row.get("MedianNs"):      # the profiled run's launches include the clock ramp of a fresh process: its median is the steady state
                                frac_rocprof_median = round(alg_bytes / (float(row["MedianNs"]) * 1e-9) / 1e9 / 8000.0, 5)
                                rocprof_src += f", median {float(row['MedianNs']) / 1e6:.4f} ms (the average includes the first launches at idle clocks)"
                    break
        except Exception:
            frac_rocprof, rocprof_src = None, None
        result["roofline"] = {"bound": "hbm", "kernel": "k_accumulate", "achieved": round(achieved, 2), "peak": 8000.0,
                              "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "frac_rocprof": frac_rocprof, "frac_rocprof_median": frac_rocprof_median, "frac_rocprof_source": rocprof_src,
                              "traffic": traffic, "traffic_source": traffic_source,
                              "avg_launch_ms": round(t_acc, 4),
                              "avg_launch_how": f"median over {acc_calls} prepared MSMs back to back (the headline's state), in-library HIP events on the launch stream appended call after call and read back once",
                              "algorithmic_bytes_per_launch": alg_bytes,
                              "whole_msm_frac": round(alg_bytes / (ms_per_step * 1e-3) / 1e9 / 8000.0, 5),
                              "whole_msm_traffic": whole_msm_traffic,
                              "note": "256-bit modular-integer work: ALU-bound, not HBM-bound (DESIGN.md)"}
        # issue-rate view of the same kernel: 8M + 2S mixed addition with Y3's two products under one reduction
        # = 6*162 + (2*81 + 81) + 2*126 = 1467 v_mad_u64_u32 (ISA count of the loop body),
        # W*n additions per launch (W = ceil(256 / c) windows of the prepared table); peak = the v_mad_u64_u32-only issue rate measured by tools/isa_rate.hip at 4 waves/SIMD
        c_bits = lib.zkhip_prepared_window_bits(handle)
        windows = (256 + c_bits - 1) // c_bits
        imads = 1467.0 * float(windows) * n
        result["valu_roofline"] = {"kernel": "k_accumulate", "window_bits": c_bits, "windows": windows, "unit": "T v_mad_u64_u32 lane-ops/s", "achieved": round(imads / (t_acc * 1e-3) / 1e12, 2),
                                   "peak": 27.95, "frac": round(imads / (t_acc * 1e-3) / 1e12 / 27.95, 3),
                                   "note": "the other 31% of the loop body's 2130 instructions (carry shifts, masks, limb adds) share the same issue slots"}
        result["phases_ms"] = {k: round(v, 4) for k, v in acc.items()}
        # the phases partition one call: their sum must not exceed the step (the events add a little; 5 % allowed).  A failure marks this extra, not the line.
        ph_sum = sum(acc.values())
        result["phases_check"] = {"sum_ms": round(ph_sum, 4), "ms_per_step": round(ms_per_step, 4), "ok": bool(ph_sum <= 1.05 * ms_per_step and not exchange)}
        if exchange:
            result["phases_check"]["note"] = "the step holds the exchange as well: not comparable"

    # ---- BASELINE configs[4]: the wrapper circuit at k + 2 (2^24 points) sharded over the GPUs of the node -------------------------
    # N > 1: every rank takes 2^24 / N points (2^21 at N = 8) -- the stated config, next to the weak-scaling headline above.
    # N = 1: the same 2^24-point MSM through the C ABI's own multi-GPU path with 8 virtual shards of 2^21 points on the one card.
    if not args.no_extras:
        try:
            c4 = config4_leg(lib, _lib, F, torch, dist, dev, stream, rank, world, gather_fold_device, cpu_group=cpu_group)
            if rank == 0:
                result["config4_wrapper_k24_msm"] = c4
        except Exception as exc:   # an extra: never fail the bench line
            if rank == 0:
                result["config4_wrapper_k24_msm"] = {"error": repr(exc)}

    # N > 1, last: the SAME configs[4] MSM driven by ONE process over all N cards through the C ABI (zkhip_init([0..N-1]) + registered shards) --
    # the path a Rust host would use -- on rank 0, while the other ranks have released their buffers and wait on the host (gloo) so that
    # their cards are idle.  Re-initialising the library drops rank 0's earlier handles: nothing below needs them.
    if world > 1 and not args.no_extras:
        del d_bases, d_scalars
        torch.cuda.synchronize()
        if rank != 0:
            lib.zkhip_shutdown()
        torch.cuda.empty_cache()
        dist.barrier(group=cpu_group)
        if rank == 0:
            # in a CHILD process: one process driving several cards (peer access, peer copies, cross-device events) has only ever run as several
            # contexts on one card -- a fault there must not take this process, and with it the N-GPU line, down
            try:
                import subprocess

                lib.zkhip_shutdown()
                torch.cuda.empty_cache()
                env = {k_: v_ for k_, v_ in os.environ.items() if k_ not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                                             "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
                child = subprocess.run([sys.executable, os.path.abspath(__file__), "--c-abi-config4", str(world)], capture_output=True, text=True, timeout=600, env=env)
                last = [ln for ln in child.stdout.splitlines() if ln.startswith("{")]
                result["config4_wrapper_k24_msm"]["single_process_c_abi"] = (json.loads(last[-1]) if child.returncode == 0 and last else
                                                                             {"error": f"child exited with {child.returncode}: {(child.stdout + child.stderr)[-300:]}"})
            except Exception as exc:   # an extra: never fail the bench line
                result["config4_wrapper_k24_msm"]["single_process_c_abi"] = {"error": repr(exc)}
        dist.barrier(group=cpu_group)

    if rank == 0 and world == 1 and not args.no_extras:
        result.update(extras(lib, _lib, F, torch, dev, stream))
        # the wrapper-size MSM (BASELINE configs[3]: k = 22, /root/reference/aggregator/benches/wrapper_circuit.rs:21) beside the headline
        if "msm_2^22" in result:
            result["value_2p22"] = result["msm_2^22"]["Mpoints_per_s"]
        try:
            result["wrapper_replay_k24"] = wrapper_replay_k24(lib, _lib, F, torch, dev, stream, result.get("config4_wrapper_k24_msm", {}))
        except Exception as exc:   # an extra: never fail the bench line
            result["wrapper_replay_k24"] = {"error": repr(exc)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.log_n, d_scalars, d_bases, d_out, n)

    if rank == 0:
        if use_dist and world == 1:
            result["rccl_rehearsal"] = True
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.destroy_process_group()


CONFIG4_T0, CONFIG4_D = 0x5A4B534E41500002 + 424242, 0x9E3779B97F4A7C15F39CC0605CEDC835


def config4_leg(lib, _lib, F, torch, dist, dev, stream, rank, world, gather_fold_device, debug: dict = None, cpu_group=None) -> dict:
    """BASELINE configs[4] (wrapper_circuit at bench-k + 2: one 2^24-point MSM over the node's GPUs, partial sums exchanged and folded).
    world > 1: strong-scaling leg -- rank g holds points [g n / N, (g + 1) n / N) of the 2^24 (device-resident, prepared), the step is its
    MSM + the all_gather(96 B) + fold, timed like the headline (barrier + synchronize on both sides, max over ranks).
    world == 1: the C ABI's multi-GPU path rehearsed on one card: zkhip_set_msm_shards(8), zkhip_register_bases on the 2^24-point SRS
    (eight tables of 2^21 points), zkhip_msm_g1 with host scalars (PCIe-inclusive: this is the host-buffer boundary) and, for the
    kernel-side number, the eight shard MSMs + fold device-resident."""
    total = 1 << 24
    per = total // world
    T0, D = CONFIG4_T0, CONFIG4_D
    dd = F.fr_encode([D])[0]
    steps = 5
    if world > 1:
        t0 = F.fr_encode([(T0 + rank * per * D) % F.R_MOD])[0]
        g = torch.empty(per * 8, dtype=torch.int64, device=dev)
        _lib.check(lib.zkhip_g1_gen_walk_device(t0.ctypes.data, dd.ctypes.data, per, g.data_ptr(), stream))
        torch.cuda.synchronize()
        h = C.c_uint64(0)
        _lib.check(lib.zkhip_prepare_bases_device(g.data_ptr(), per, C.byref(h)))
        h_sc = synth_scalars(per, 0x5A4B534E41500103 + rank)
        sc = torch.from_numpy(h_sc.view(np.int64)).to(dev)
        part = torch.zeros(12, dtype=torch.int64, device=dev)
        gat = torch.zeros(12 * world, dtype=torch.int64, device=dev)
        fin = torch.zeros(12, dtype=torch.int64, device=dev)

        def step():
            _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, sc.data_ptr(), per, part.data_ptr(), stream))
            gather_fold_device(part, gat, fin, stream)

        step()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        el = torch.tensor([time.perf_counter() - t], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        ms = float(el.item()) / steps * 1e3
        # every rank's partial against the identity of its slice of the walk, the gathered slots and the fold (see verify_exchange); the devices
        # were checked by the headline leg
        ver = verify_exchange(lib, _lib, F, torch, dist, dev, stream, cpu_group, world, sc, per, (T0 + rank * per * D) % F.R_MOD, D, part, gat, fin, check_devices=False)
        lib.zkhip_release_bases(h)
        if debug is not None:       # the rehearsal test checks the folded point against the structured identity
            debug.update(final=fin.cpu().numpy().view(np.uint64)[:12].copy(), scalars=h_sc, t0=(T0 + rank * per * D) % F.R_MOD, d=D)
        return {"workload": f"BASELINE configs[4]: 2^24-point MSM, 2^{per.bit_length() - 1} points per GPU x {world} GPUs, all_gather(96 B) + fold, device-resident",
                "scaling": "strong", "ms_per_msm": round(ms, 4), "Mpoints_per_s": round(total / ms / 1e3, 1), "steps": steps,
                "ranks_seen": ver["ranks_seen"], "result_checked": ver.get("result_checked")}
    # one card: 8 virtual shards through the C ABI
    return c_abi_config4(lib, _lib, F, torch, dev, stream, devices=[dev.index or 0], shards=8)


def c_abi_config4(lib, _lib, F, torch, dev, stream, devices, shards) -> dict:
    """BASELINE configs[4] the way a single Rust `create_proof` process would drive it (/root/reference/aggregator/src/wrapper.rs:129):
    ONE process, zkhip_init(devices), zkhip_register_bases on the 2^24-point SRS (one prepared table of 2^24 / shards points per shard, shard s
    on devices[s % ndev]), then the commit two ways -- zkhip_msm_g1 with host scalars (PCIe-inclusive: the plain two-function boundary) and
    zkhip_msm_g1_registered_device with the scalars resident in the primary device's HBM (each other device pulls its slice over xGMI, runs its
    shard, returns 96 bytes; fold on the primary).  One card: `shards` virtual shards; N cards: one shard per card."""
    total = 1 << 24
    T0, D = CONFIG4_T0, CONFIG4_D
    dd = F.fr_encode([D])[0]
    ndev = len(devices)
    devs = (C.c_int * ndev)(*devices)
    _lib.check(lib.zkhip_init(devs, ndev))          # a different device list shuts the library down first (handles of earlier legs are gone)
    g = torch.empty(total * 8, dtype=torch.int64, device=dev)
    t0 = F.fr_encode([T0])[0]
    _lib.check(lib.zkhip_g1_gen_walk_device(t0.ctypes.data, dd.ctypes.data, total, g.data_ptr(), stream))
    torch.cuda.synchronize()
    h_g = g.cpu().numpy().view(np.uint64).reshape(total, 8).copy()
    del g
    torch.cuda.empty_cache()
    h_sc = synth_scalars(total, 0x5A4B534E41500103)
    out = np.zeros(12, dtype=np.uint64)
    res = {"workload": f"BASELINE configs[4] through the C ABI of one process: 2^24-point MSM as {shards} shards of 2^{(total // shards).bit_length() - 1} points on "
                       f"{ndev} device(s) (zkhip_init + zkhip_set_msm_shards({shards}) + zkhip_register_bases: per-shard tables, per-shard Pippenger, "
                       "gather of the 96-byte partials on the primary device, fold)",
           "devices": list(devices), "shards": shards}
    _lib.check(lib.zkhip_set_msm_shards(shards))
    try:
        t = time.perf_counter()
        _lib.check(lib.zkhip_register_bases(h_g.ctypes.data, total))
        res["register_bases_s_one_time"] = round(time.perf_counter() - t, 3)
        _lib.check(lib.zkhip_msm_g1(h_sc.ctypes.data, h_g.ctypes.data, total, out.ctypes.data))
        t = time.perf_counter()
        for _ in range(3):
            _lib.check(lib.zkhip_msm_g1(h_sc.ctypes.data, h_g.ctypes.data, total, out.ctypes.data))
        ms_host = (time.perf_counter() - t) / 3 * 1e3
        res["host_buffers_ms_per_msm"] = round(ms_host, 2)        # scalars cross PCIe (512 MiB per call)
        res["host_buffers_Mpoints_per_s"] = round(total / ms_host / 1e3, 1)
        # device-resident scalars (primary device) against the same registered shards
        sc = torch.from_numpy(h_sc.view(np.int64)).to(dev)
        fin = torch.zeros(12, dtype=torch.int64, device=dev)
        run = lambda: _lib.check(lib.zkhip_msm_g1_registered_device(h_g.ctypes.data, sc.data_ptr(), total, fin.data_ptr(), stream))
        run()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 3 * 1e3
        same = F.g1_decode_jacobian(fin.cpu().numpy().view(np.uint64)[:12]) == F.g1_decode_jacobian(out)
        res.update({"device_resident_c_abi_ms": round(ms, 3), "device_resident_ms_per_msm": round(ms, 3), "device_resident_Mpoints_per_s": round(total / ms / 1e3, 1),
                    "host_and_device_paths_agree": bool(same)})
        if ndev == 1:
            res["per_shard_ms"] = round(ms / shards, 3)
            res["note"] = "one card: the shards run one after the other; on N cards they run concurrently (expected ~ per_shard_ms x shards / N + the slice copy and the exchange)"
        else:
            res["note"] = "N cards, one process: secondary devices copy their scalar slice from the primary's HBM (hipMemcpyPeerAsync), partials return as 96-byte peer copies"
        del sc
    finally:
        lib.zkhip_unregister_bases(h_g.ctypes.data)
        lib.zkhip_set_msm_shards(0)
    del h_g
    torch.cuda.empty_cache()
    return res


def extras(lib, _lib, F, torch, dev, stream) -> dict:
    """NTT throughput at the wrapper sizes and the MSM+NTT call mix of one wrapper-circuit proof (SURVEY.md 3.2 / 8d
    config 4: 18 MSM 2^22 + 13 iNTT 2^22 + 13 NTT 2^24 + 1 iNTT 2^24), everything device-resident."""
    out = {}
    from zksnap_circuits_halo2_amd.fields import R_MOD, omega_for

    def timed(fn, reps):
        prewarm(fn, torch, 1, PREWARM_MS / 2)          # (at least one call; the clocks are up when the timed calls start)
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3

    ntt = {}
    bufs = {}
    for L in (22, 24):
        N = 1 << L
        a = torch.from_numpy(synth_scalars(N, 1000 + L).view(np.int64)).to(dev)
        bufs[L] = a
        om = F.fr_encode([omega_for(L)])[0]
        ms = timed(lambda: _lib.check(lib.zkhip_ntt_fr_device(a.data_ptr(), om.ctypes.data, L, stream)), 5)
        ntt[f"2^{L}"] = {"ms": round(ms, 4), "Melem_per_s": round(N / ms / 1e3, 1),
                         "hbm_frac_algorithmic": round(64.0 * N / (ms * 1e-3) / 8e12, 5)}
        # per-pass durations of the same transform: HIP events the library drops on the launch stream between its passes
        lib.zkhip_profile_enable(1)
        per_pass = {}
        for _ in range(3):
            _lib.check(lib.zkhip_ntt_fr_device(a.data_ptr(), om.ctypes.data, L, stream))
            for name, t_ms in profile_read(lib):
                per_pass[name] = per_pass.get(name, 0.0) + t_ms / 3
        lib.zkhip_profile_enable(0)
        ntt[f"2^{L}"]["passes_ms"] = {kk: round(v, 4) for kk, v in per_pass.items()}
    out["ntt"] = ntt
    # roofline object of the NTT's kernel (64 B per element per transform, one 32-byte read and one 32-byte write;
    # a pass moves N x 32 B in and N x 32 B out, PMC-confirmed, so a transform of p passes has p x 64 B per element of traffic)
    p24 = ntt["2^24"]["passes_ms"]
    ntt_traffic, ntt_traffic_source = None, None
    try:   # PMC bytes per k_ntt_pass launch at 2^24: separate rocprofv3 --pmc passes (tools/collect_profiles.sh), a committed constant like roofline.traffic
        path = os.path.join(ROOT, "profiles", "r05_pmc_traffic_ntt.json")
        if not os.path.exists(path):
            path = os.path.join(ROOT, "profiles", "r04_pmc_traffic_ntt.json")
        if os.path.exists(path):
            rec = json.load(open(path))
            rows = [v for k_, v in rec["ntt"]["2^24"].items() if "k_ntt_pass" in k_]
            if rows:
                tot_d = sum(v["dispatches"] for v in rows)
                ntt_traffic = round(sum(v["hbm_bytes_per_launch"] * v["dispatches"] for v in rows) / tot_d)
                ntt_traffic_source = f"profiles/{os.path.basename(path)}@{rec.get('commit', '?')}: average over the k_ntt_pass launches of the 2^24 transform (FETCH_SIZE x2 + WRITE_SIZE)"
    except Exception:
        ntt_traffic, ntt_traffic_source = None, None
    if p24:
        avg_pass = sum(p24.values()) / len(p24)
        out["roofline_ntt"] = {"bound": "hbm", "kernel": "k_ntt_pass", "workload": "NTT 2^24 (three passes)", "avg_pass_ms": round(avg_pass, 4),
                               "algorithmic_bytes_per_launch": 64.0 * (1 << 24), "achieved": round(64.0 * (1 << 24) / (avg_pass * 1e-3) / 1e9, 1),
                               "peak": 8000.0, "unit": "GB/s", "frac": round(64.0 * (1 << 24) / (avg_pass * 1e-3) / 1e9 / 8000.0, 4),
                               "whole_transform_frac": ntt["2^24"]["hbm_frac_algorithmic"],
                               "traffic": ntt_traffic, "traffic_source": ntt_traffic_source,
                               "note": "per launch = one pass over the 2^24 elements (N x 32 B read + N x 32 B written); the kernel is bound by VALU issue (254-bit modular multiplies), DESIGN.md section 4"}

    k = 22
    n = 1 << k
    t0 = F.fr_encode([12345])[0]
    dd = F.fr_encode([0x9E3779B97F4A7C15])[0]
    g = torch.empty(n * 8, dtype=torch.int64, device=dev)
    _lib.check(lib.zkhip_g1_gen_walk_device(t0.ctypes.data, dd.ctypes.data, n, g.data_ptr(), stream))
    sc = bufs[22]
    res = torch.zeros(16, dtype=torch.int64, device=dev)
    h22 = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(g.data_ptr(), n, C.byref(h22)))
    om22i = F.fr_encode([pow(omega_for(22), -1, R_MOD)])[0]
    div22 = F.fr_encode([pow(n, -1, R_MOD)])[0]
    om24 = F.fr_encode([omega_for(24)])[0]
    om24i = F.fr_encode([pow(omega_for(24), -1, R_MOD)])[0]
    div24 = F.fr_encode([pow(1 << 24, -1, R_MOD)])[0]
    ext = bufs[24]

    def replay():
        for _ in range(18):
            _lib.check(lib.zkhip_msm_g1_prepared_device(h22, 0, sc.data_ptr(), n, res.data_ptr(), stream))
        for _ in range(13):
            _lib.check(lib.zkhip_ifft_scaled_device(sc.data_ptr(), om22i.ctypes.data, 22, div22.ctypes.data, stream))
        for _ in range(13):
            _lib.check(lib.zkhip_ntt_fr_device(ext.data_ptr(), om24.ctypes.data, 24, stream))
        _lib.check(lib.zkhip_ifft_scaled_device(ext.data_ptr(), om24i.ctypes.data, 24, div24.ctypes.data, stream))

    ms_msm22 = timed(lambda: _lib.check(lib.zkhip_msm_g1_prepared_device(h22, 0, sc.data_ptr(), n, res.data_ptr(), stream)), 3)
    ms = timed(replay, 2)
    # the same mix with the 18 independent commits alternating between two streams (each has its own scratch set inside the library):
    # one MSM's latency-bound reduction tail runs under the next one's accumulation
    ms_two_streams = None
    ms_msm_beside_ntt = None
    try:
        # (the current stream and a high-priority one: streams of different priority never share a hardware queue)
        side = [torch.cuda.current_stream(), torch.cuda.Stream(priority=-1)]
        res2 = [torch.zeros(16, dtype=torch.int64, device=dev) for _ in range(2)]

        def replay2():
            cur = torch.cuda.current_stream()
            side[1].wait_stream(cur)
            for i in range(18):
                _lib.check(lib.zkhip_msm_g1_prepared_device(h22, 0, sc.data_ptr(), n, res2[i % 2].data_ptr(), C.c_void_p(side[i % 2].cuda_stream)))
            cur.wait_stream(side[1])
            for _ in range(13):
                _lib.check(lib.zkhip_ifft_scaled_device(sc.data_ptr(), om22i.ctypes.data, 22, div22.ctypes.data, stream))
            for _ in range(13):
                _lib.check(lib.zkhip_ntt_fr_device(ext.data_ptr(), om24.ctypes.data, 24, stream))
            _lib.check(lib.zkhip_ifft_scaled_device(ext.data_ptr(), om24i.ctypes.data, 24, div24.ctypes.data, stream))

        ms_two_streams = timed(replay2, 2)

        def replay3():         # the commits on the side stream, the transforms on the current one (a column's commit and its extended transform both
            cur = torch.cuda.current_stream()      # read its coefficients and are independent of each other)
            side[1].wait_stream(cur)
            for i in range(18):
                _lib.check(lib.zkhip_msm_g1_prepared_device(h22, 0, sc2.data_ptr(), n, res2[1].data_ptr(), C.c_void_p(side[1].cuda_stream)))
            for _ in range(13):
                _lib.check(lib.zkhip_ifft_scaled_device(sc.data_ptr(), om22i.ctypes.data, 22, div22.ctypes.data, stream))
            for _ in range(13):
                _lib.check(lib.zkhip_ntt_fr_device(ext.data_ptr(), om24.ctypes.data, 24, stream))
            _lib.check(lib.zkhip_ifft_scaled_device(ext.data_ptr(), om24i.ctypes.data, 24, div24.ctypes.data, stream))
            cur.wait_stream(side[1])

        sc2 = sc.clone()       # the MSMs read a copy: the transforms run in place on `sc`
        ms_msm_beside_ntt = timed(replay3, 2)
        del sc2
        # the same mix with the 13 extended transforms issued the way the device-resident prover issues them: `coeff_to_extended` as ONE fused
        # call (coset scale + zero padding + transform: the first pass reads n coefficients, not 4n) instead of a full in-place 2^24 transform
        try:
            from zksnap_circuits_halo2_amd.domain import EvaluationDomain
            dom22 = EvaluationDomain(4, 22)

            def replay4():
                for _ in range(18):
                    _lib.check(lib.zkhip_msm_g1_prepared_device(h22, 0, sc.data_ptr(), n, res.data_ptr(), stream))
                for _ in range(13):
                    _lib.check(lib.zkhip_ifft_scaled_device(sc.data_ptr(), om22i.ctypes.data, 22, div22.ctypes.data, stream))
                for _ in range(13):
                    _lib.check(lib.zkhip_coeff_to_extended_device(sc.data_ptr(), n, 22, ext.data_ptr(), 1 << 24, 24, 1, dom22.extended_omega.ctypes.data,
                                                                  dom22.g_coset.ctypes.data, stream))
                _lib.check(lib.zkhip_ifft_scaled_device(ext.data_ptr(), om24i.ctypes.data, 24, div24.ctypes.data, stream))

            ms_fused_ext = timed(replay4, 2)
        except Exception as exc:
            ms_fused_ext = repr(exc)
    except Exception as exc:   # an extra: never fail the bench line
        ms_two_streams = repr(exc)
        ms_msm_beside_ntt = None
        ms_fused_ext = None
    # the same 2^22 MSM under scalar distributions that real columns have: buckets that hold a large share of all entries must not
    # serialise anything (profiles/r02_scalar_distributions.txt, tools/skew_probe.py)
    try:
        one_w = torch.from_numpy(F.fr_encode([1])[0].view(np.int64)).to(dev)
        neg_w = torch.from_numpy(F.fr_encode([R_MOD - 1])[0].view(np.int64)).to(dev)
        keep = sc.clone()
        dist = {}
        def t_sc():
            return round(timed(lambda: _lib.check(lib.zkhip_msm_g1_prepared_device(h22, 0, sc.data_ptr(), n, res.data_ptr(), stream)), 3), 3)
        sc[::20] = neg_w
        dist["uniform_5pct_equal_to_minus_one_ms"] = t_sc()
        sc.zero_(); sc[::2] = one_w
        dist["selector_half_ones_ms"] = t_sc()
        sc[:] = one_w
        dist["all_ones_ms"] = t_sc()
        sc.copy_(keep)
        del keep
        dist["uniform_ms"] = round(ms_msm22, 3)
        out["msm_2^22_scalar_distributions"] = dist
    except Exception as exc:   # an extra: never fail the bench line
        out["msm_2^22_scalar_distributions"] = {"error": repr(exc)}
    lib.zkhip_release_bases(h22)
    # the same call mix through the host-buffer entry points (what the plain two-function drop-in sees: every scalar vector and
    # every polynomial crosses PCIe from / to pageable host memory; the SRS is registered once, as INTEGRATION.md describes)
    h_g = g.cpu().numpy().view(np.uint64).reshape(n, 8).copy()
    h_sc = sc.cpu().numpy().view(np.uint64).reshape(n, 4).copy()
    h_ext = ext.cpu().numpy().view(np.uint64).reshape(1 << 24, 4).copy()
    h_out = np.zeros(12, dtype=np.uint64)
    _lib.check(lib.zkhip_register_bases(h_g.ctypes.data, n))

    def replay_host():
        for _ in range(18):
            _lib.check(lib.zkhip_msm_g1(h_sc.ctypes.data, h_g.ctypes.data, n, h_out.ctypes.data))
        for _ in range(13):
            _lib.check(lib.zkhip_ifft_scaled(h_sc.ctypes.data, om22i.ctypes.data, 22, div22.ctypes.data))
        for _ in range(13):
            _lib.check(lib.zkhip_ntt_fr(h_ext.ctypes.data, om24.ctypes.data, 24))
        _lib.check(lib.zkhip_ifft_scaled(h_ext.ctypes.data, om24i.ctypes.data, 24, div24.ctypes.data))

    replay_host()
    t_h = time.perf_counter()
    replay_host()
    ms_host = (time.perf_counter() - t_h) * 1e3
    # the same mix with the shim's optional domain_patch.rs applied (rust-shim/): `coeff_to_extended` is ONE fused call that uploads the 2^22
    # coefficients only (128 MiB instead of 512 MiB of mostly zeros) and downloads the 2^24 evaluations; `extended_to_coeff` downloads the 3 * 2^22
    # coefficients it keeps
    zeta_m = F.fr_encode([F.ZETA])[0]
    h_out24 = np.empty_like(h_ext)
    h_quot = np.empty((3 << 22, 4), dtype=np.uint64)

    def replay_host_fused():
        for _ in range(18):
            _lib.check(lib.zkhip_msm_g1(h_sc.ctypes.data, h_g.ctypes.data, n, h_out.ctypes.data))
        for _ in range(13):
            _lib.check(lib.zkhip_ifft_scaled(h_sc.ctypes.data, om22i.ctypes.data, 22, div22.ctypes.data))
        for _ in range(13):
            _lib.check(lib.zkhip_coeff_to_extended(h_sc.ctypes.data, 22, h_out24.ctypes.data, 24, om24.ctypes.data, zeta_m.ctypes.data))
        _lib.check(lib.zkhip_extended_to_coeff(h_ext.ctypes.data, 24, om24i.ctypes.data, div24.ctypes.data, zeta_m.ctypes.data, h_quot.ctypes.data, 3 << 22))

    replay_host_fused()
    t_h = time.perf_counter()
    replay_host_fused()
    ms_host_fused = (time.perf_counter() - t_h) * 1e3
    del h_out24, h_quot
    # the same op list issued by TWO host threads (each with its own buffers): host-buffer calls borrow separate lanes of the library, so one
    # call's PCIe transfer runs under the other's kernels and the two PCIe directions are busy at once.  The reference's prover issues these
    # calls from one thread (columns are committed / transformed one after another), so this is the ceiling a batching host could reach,
    # not the drop-in number.
    import threading

    h_sc2, h_ext2, h_out2 = h_sc.copy(), h_ext.copy(), np.zeros(12, dtype=np.uint64)

    def half(sc_, ext_, out_, n_msm, n_i22, n_f24, n_i24):
        for _ in range(n_msm):
            _lib.check(lib.zkhip_msm_g1(sc_.ctypes.data, h_g.ctypes.data, n, out_.ctypes.data))
        for _ in range(n_i22):
            _lib.check(lib.zkhip_ifft_scaled(sc_.ctypes.data, om22i.ctypes.data, 22, div22.ctypes.data))
        for _ in range(n_f24):
            _lib.check(lib.zkhip_ntt_fr(ext_.ctypes.data, om24.ctypes.data, 24))
        for _ in range(n_i24):
            _lib.check(lib.zkhip_ifft_scaled(ext_.ctypes.data, om24i.ctypes.data, 24, div24.ctypes.data))

    def replay_host_2():
        ta = threading.Thread(target=half, args=(h_sc, h_ext, h_out, 9, 7, 6, 1))
        tb = threading.Thread(target=half, args=(h_sc2, h_ext2, h_out2, 9, 6, 7, 0))
        ta.start(); tb.start(); ta.join(); tb.join()

    replay_host_2()
    t_h = time.perf_counter()
    replay_host_2()
    ms_host2 = (time.perf_counter() - t_h) * 1e3
    # single ops through the host-buffer boundary (PCIe-inclusive)
    t_h = time.perf_counter()
    for _ in range(5):
        _lib.check(lib.zkhip_msm_g1(h_sc.ctypes.data, h_g.ctypes.data, n, h_out.ctypes.data))
    ms_msm22_host = (time.perf_counter() - t_h) / 5 * 1e3
    t_h = time.perf_counter()
    for _ in range(3):
        _lib.check(lib.zkhip_ntt_fr(h_ext.ctypes.data, om24.ctypes.data, 24))
    ms_ntt24_host = (time.perf_counter() - t_h) / 3 * 1e3
    _lib.check(lib.zkhip_unregister_bases(h_g.ctypes.data))
    del h_g, h_sc, h_ext, h_sc2, h_ext2
    out["msm_2^22"] = {"ms": round(ms_msm22, 3), "Mpoints_per_s": round(n / ms_msm22 / 1e3, 1)}
    del g, ext, sc
    bufs.clear()
    torch.cuda.empty_cache()
    # G2 MSM (best_multiexp::<G2Affine>; no prover call site): 2^16 and 2^20 points = 64 distinct multiples of the generator (host big-integer
    # arithmetic of the package's srs module) repeated, uniform scalars.  Round 5: GLV digits on the twist, lazy Fq2 arithmetic in the accumulation.
    try:
        from zksnap_circuits_halo2_amd import srs as _srs

        base = np.stack([_srs.g2_encode(_srs.g2_mul(1000003 * (i + 1))) for i in range(64)])
        r_inv = pow(1 << 256, -1, R_MOD)
        mont_inv = pow(1 << 256, -1, _srs.Q_MOD)
        for log_n2 in (16, 20):
            n2 = 1 << log_n2
            d_b2 = torch.from_numpy(np.ascontiguousarray(np.tile(base, (n2 // 64, 1))).view(np.int64)).to(dev)
            d_s2 = torch.from_numpy(synth_scalars(n2, 4242 + log_n2).view(np.int64)).to(dev)
            d_o2 = torch.zeros(24, dtype=torch.int64, device=dev)
            ms_g2 = timed(lambda: _lib.check(lib.zkhip_msm_g2_device(d_s2.data_ptr(), d_b2.data_ptr(), n2, d_o2.data_ptr(), stream)), 3)
            # the result is checked, not discarded: base i = (1000003 (i % 64 + 1)) G2, so MSM = [sum a_i 1000003 (i % 64 + 1)] G2 (host integers of the
            # package's srs module; the scalars' integer values a = words / 2^256 mod r, summed per residue class of i mod 64 with numpy object arrays)
            h_s2 = d_s2.cpu().numpy().view(np.uint64).reshape(n2, 4)
            col = [h_s2[:, j].astype(object) for j in range(4)]
            vals = col[0] + (col[1] << 64) + (col[2] << 128) + (col[3] << 192)
            want_k = 0
            for c_ in range(64):
                want_k = (want_k + int(vals[c_::64].sum()) % R_MOD * r_inv % R_MOD * 1000003 * (c_ + 1)) % R_MOD
            jac = d_o2.cpu().numpy().view(np.uint64)[:24]
            v = [sum(int(jac[4 * c_ + j]) << (64 * j) for j in range(4)) * mont_inv % _srs.Q_MOD for c_ in range(6)]
            X, Y, Zc = (v[0], v[1]), (v[2], v[3]), (v[4], v[5])
            iz = _srs._f2inv(Zc)
            iz2 = _srs._f2mul(iz, iz)
            got_pt = (_srs._f2mul(X, iz2), _srs._f2mul(Y, _srs._f2mul(iz2, iz)))
            out["msm_g2_2^%d" % log_n2] = {"ms": round(ms_g2, 3), "Mpoints_per_s": round(n2 / ms_g2 / 1e3, 2),
                                           "result_equals_structured_identity": bool(got_pt == _srs.g2_mul(want_k)),
                                           "note": "general path (GLV digits, c = 16); not on the prover's path"}
            del d_b2, d_s2
    except Exception as exc:   # an extra: never fail the bench line
        out["msm_g2_2^16"] = {"error": repr(exc)}
    out["small_circuit_replays"] = small_replays(lib, _lib, F, torch, dev, stream, timed)
    out["small_circuit_replays"]["prover_sequences_c99"] = shim_sequences_small(torch)
    out["prover_phases_k22"] = prover_phases(lib, _lib, F, torch, dev, stream, timed)
    out["wrapper_replay"] = {"workload": "k=22: 18 MSM 2^22 + 13 iNTT 2^22 + 13 NTT 2^24 + 1 iNTT 2^24, device-resident",
                             "ms": round(ms, 2), "proofs_per_s_msm_ntt_portion": round(1e3 / ms, 3),
                             "note": "MSM+NTT portion only; the Rust host (witness, transcript) cannot run here"}
    out["wrapper_replay"]["ms_commits_on_two_streams"] = round(ms_two_streams, 2) if isinstance(ms_two_streams, float) else ms_two_streams
    out["wrapper_replay"]["ms_commits_beside_transforms"] = round(ms_msm_beside_ntt, 2) if isinstance(ms_msm_beside_ntt, float) else ms_msm_beside_ntt
    out["wrapper_replay"]["ms_with_fused_coeff_to_extended"] = round(ms_fused_ext, 2) if isinstance(ms_fused_ext, float) else ms_fused_ext   # 13 x zkhip_coeff_to_extended_device (2^22 -> 2^24) in place of 13 full 2^24 transforms
    out["wrapper_replay"]["host_buffers_ms"] = round(ms_host, 1)              # PCIe-inclusive: never `value`
    out["wrapper_replay"]["proofs_per_s_host_buffers"] = round(1e3 / ms_host, 3)
    out["wrapper_replay"]["host_buffers_with_fused_domain_calls_ms"] = round(ms_host_fused, 1)    # rust-shim/domain_patch.rs applied
    out["wrapper_replay"]["host_buffers_two_caller_threads_ms"] = round(ms_host2, 1)
    out["wrapper_replay"]["host_buffers_single_ops_ms"] = {"msm_2^22_registered_bases": round(ms_msm22_host, 2), "ntt_2^24": round(ms_ntt24_host, 2),
                                                           "note": "pageable host memory both ways; the NTT moves 512 MiB each way (~19 ms of PCIe at ~55 GB/s) around a 2.4 ms kernel"}
    # a transcript-less create_proof of a *satisfied* halo2-lib-shaped circuit (4 gate columns, lookup, copy constraints), device-resident,
    # with the prover's invariants checked (tools/prove_flow.py): quotient is a polynomial, grand products close, commitments agree
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import prove_flow

        flows = [prove_flow.run(22, 4, seed=22 + r, verbose=False) for r in range(2)]     # the first run pays one-time allocations
        assert all(all(f["checks"].values()) for f in flows), [f["checks"] for f in flows]
        flow = min(flows, key=lambda f: f["prove_ms"])
        out["prover_flow_k22"] = {"prove_ms": round(flow["prove_ms"], 2), "columns": flow["columns"], "msms": flow["msms"], "checks": flow["checks"],
                                  "timings_ms": {kk: round(v, 2) for kk, v in flow["timings_ms"].items()},
                                  "proof_columns": flow["proof_columns"],
                                  "note": "no transcript (seeded challenges); the proving key's columns (fixed, l_0/l_last/l_active, sigma) are transformed under keygen_* and not counted, as pk.fixed_cosets / pk.permutation.cosets are in the reference"}
    except Exception as exc:   # an extra: never fail the bench line
        out["prover_flow_k22"] = {"error": repr(exc)}
    # the same flow at the small circuits' real width (columns sized by calculate_params(Some(20)): /root/reference/voter/benches/voter_circuit.rs:49-51,
    # /root/reference/aggregator/benches/state_transition_circuit.rs:48-50): voter-like k = 13 with 256 gate + 8 lookup columns, state-transition-like
    # k = 15 with 64 + 8; commits batched per phase; quotient, grand products, lookup permutations and the SHPLONK multi-open included
    for name, kk, gg, ll in (("prover_flow_k13_wide", 13, 256, 8), ("prover_flow_k15_wide", 15, 64, 8)):
        try:
            flows = [prove_flow.run(kk, gg, seed=kk + r, verbose=False, lookups=ll) for r in range(2)]
            assert all(all(f["checks"].values()) for f in flows), [f["checks"] for f in flows]
            flow = min(flows, key=lambda f: f["prove_ms"])
            out[name] = {"prove_ms": round(flow["prove_ms"], 2), "proofs_per_s_device_portion": round(1e3 / flow["prove_ms"], 2), "gate_columns": gg, "lookup_columns": ll,
                         "columns": flow["columns"], "msms": flow["msms"], "multiopen_queries": flow["queries"], "quotient_program_insns": flow["program_insns"],
                         "checks": flow["checks"], "timings_ms": {kk_: round(v, 2) for kk_, v in flow["timings_ms"].items()},
                         "note": "no transcript, no witness generation; keygen_* not counted (the key is per circuit)"}
        except Exception as exc:   # an extra: never fail the bench line
            out[name] = {"error": repr(exc)}
    tot = ms + out["prover_phases_k22"]["total_ms"]
    out["wrapper_replay"]["with_prover_phases_ms"] = round(tot, 2)          # + quotient, grand products, multiopen (8(f) rows 1-3)
    out["wrapper_replay"]["proofs_per_s_device_portion"] = round(1e3 / tot, 3)
    out["wrapper_replay"].update(shim_device_sequence(22, 4, torch))
    return out


def shim_device_sequence(k: int, gate_cols: int, torch) -> dict:
    """The device-resident call sequence of rust-shim/prover_patch.rs (mode (b)) for one proof of the wrapper's shape (k = 22, 4 gate columns + 1
    lookup column: /root/reference/aggregator/benches/wrapper_circuit.rs:61-68), issued by the C99 replay tests/cpp/prover_sequence.c in a child
    process: witness upload, advice / lookup / product commitments, grand products, lagrange_to_coeff, coeff_to_extended, the quotient program,
    its commitments and the evaluations at x, with the commitments and evaluations read back (what the transcript needs); then (reported
    separately) the multi-open argument -- SHPLONK over the same device-resident polynomials.  PCIe-inclusive."""
    import re
    import subprocess
    import tempfile

    try:
        from zksnap_circuits_halo2_amd import evaluation as E

        lib_dir = os.path.join(ROOT, "zksnap_circuits_halo2_amd")
        with tempfile.TemporaryDirectory() as tmp:
            exe, rec = os.path.join(tmp, "prover_sequence"), os.path.join(tmp, "programs.bin")
            subprocess.check_call(["gcc", "-std=c99", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "prover_sequence.c"),
                                   "-o", exe, "-L", lib_dir, "-lzkhip", "-Wl,-rpath," + lib_dir])
            with open(rec, "wb") as fh:
                fh.write(E.export_prover_programs(k, gate_cols, 1, seed=k))
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            res = subprocess.run([exe, rec, "--device-only", "3"], capture_output=True, text=True, timeout=600)
        m = re.search(r"sequence_ms device_resident=([\d.]+)", res.stdout)
        if res.returncode != 0 or not m:
            return {"shim_device_sequence_ms": None, "shim_device_sequence_error": (res.stdout + res.stderr)[-400:]}
        ms = float(m.group(1))
        m2 = re.search(r"with_multiopen=([\d.]+)", res.stdout)
        return {"shim_device_sequence_ms": round(ms, 2),
                "shim_device_sequence_with_multiopen_ms": round(float(m2.group(1)), 2) if m2 else None,    # + SHPLONK on the device-resident polynomials (zkhip_multiopen_shplonk_*: 2 more MSMs)
                "shim_device_sequence": {"how": "tests/cpp/prover_sequence.c --device-only 3 in a child process: the call sequence of rust-shim/prover_patch.rs mode (b), "
                                                "three proofs over the same buffers, the fastest reported (first_ms = the process's first proof, cold clocks)",
                                         "first_ms": float(re.search(r"first=([\d.]+)", res.stdout).group(1)),
                                         "shape": re.search(r"shape: (.*)", res.stdout).group(1) if "shape: " in res.stdout else None,
                                         "proofs_per_s": round(1e3 / ms, 3)}}
    except Exception as exc:   # an extra: never fail the bench line
        return {"shim_device_sequence_ms": None, "shim_device_sequence_error": repr(exc)}


def shim_sequences_small(torch) -> dict:
    """The same C99 replay at the reference's small circuits' sizes and column counts -- the voter at k = 13 and the state transition at k = 15
    size their columns with `calculate_params` (/root/reference/voter/benches/voter_circuit.rs:49-51,
    /root/reference/aggregator/benches/state_transition_circuit.rs:48-50: hundreds of advice columns at these k; BASELINE configs[0] and [2]) --
    all three ways over one witness with every commitment, evaluation and quotient coefficient compared by the program itself: host buffers
    call by call (the plain drop-in), host buffers with one call per phase (prover_patch.rs mode (a)), device-resident handles (mode (b)),
    and the last one with the multi-open argument (SHPLONK) on top.  One lookup column (the program's limit), so 256 / 64 gate columns."""
    import re
    import subprocess
    import tempfile

    out = {}
    try:
        from zksnap_circuits_halo2_amd import evaluation as E

        lib_dir = os.path.join(ROOT, "zksnap_circuits_halo2_amd")
        with tempfile.TemporaryDirectory() as tmp:
            exe = os.path.join(tmp, "prover_sequence")
            subprocess.check_call(["gcc", "-std=c99", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "prover_sequence.c"),
                                   "-o", exe, "-L", lib_dir, "-lzkhip", "-Wl,-rpath," + lib_dir])
            for name, k, gate_cols in (("voter_shape_k13_256_columns", 13, 256), ("state_transition_shape_k15_64_columns", 15, 64)):
                rec = os.path.join(tmp, f"programs_{k}.bin")
                with open(rec, "wb") as fh:
                    fh.write(E.export_prover_programs(k, gate_cols, 1, seed=k))
                torch.cuda.synchronize()
                res = subprocess.run([exe, rec], capture_output=True, text=True, timeout=600)
                m = re.search(r"sequence_ms host_call_by_call=([\d.]+) host_one_call_per_phase=([\d.]+) device_resident=([\d.]+) \(with the multi-open argument ([\d.]+)\)", res.stdout)
                if res.returncode != 0 or not m or "prover sequence OK" not in res.stdout:
                    out[name] = {"error": (res.stdout + res.stderr)[-300:]}
                    continue
                shape = re.search(r"shape: (.*)", res.stdout)
                out[name] = {"host_call_by_call_ms": float(m.group(1)), "host_one_call_per_phase_ms": float(m.group(2)), "device_resident_ms": float(m.group(3)),
                             "device_resident_with_multiopen_ms": float(m.group(4)),
                             "shape": shape.group(1) if shape else None, "results_compared": True}
                # steady state of the device-resident way: five proofs over the same buffers, the fastest
                res2 = subprocess.run([exe, rec, "--device-only", "5"], capture_output=True, text=True, timeout=600)
                m5 = re.search(r"sequence_ms device_resident=([\d.]+) first=[\d.]+ repeats=5 with_multiopen=([\d.]+)", res2.stdout)
                if res2.returncode == 0 and m5:
                    out[name]["device_resident_steady_ms"] = float(m5.group(1))
                    out[name]["device_resident_with_multiopen_steady_ms"] = float(m5.group(2))
                    out[name]["proofs_per_s_device_side"] = round(1e3 / float(m5.group(2)), 1)
        out["how"] = ("tests/cpp/prover_sequence.c in child processes: ONE proof per way (cold: the first calls of a process), then the device-resident way five times "
                      "over the same buffers (steady: the fastest); PCIe-inclusive; no witness generation, "
                      "no transcript; the proving key's columns are transformed outside the timed region, as pk.fixed_cosets / permutation.cosets are")
    except Exception as exc:   # an extra: never fail the bench line
        out["error"] = repr(exc)
    return out


def wrapper_replay_k24(lib, _lib, F, torch, dev, stream, c4: dict) -> dict:
    """BASELINE configs[4] as an OP MIX (SURVEY.md 8(d) Config 5: the wrapper's call mix at n = 2^24, extended 2^26;
    /root/reference/aggregator/benches/wrapper_circuit.rs:21,61-68 at k + 2): 18 MSM 2^24 + 13 iNTT 2^24 + 13 NTT 2^26 + 1 iNTT 2^26,
    device-resident on ONE card (13 GiB of prepared table, 2 GiB per extended polynomial + two scratch copies), MSM and NTT parts timed
    separately, and the 8-GPU projection: the MSMs sharded by point range (2^21 per card: `per_shard_ms` of the configs[4] leg) + the
    exchange, the transforms unsharded on the primary card (north star: NTT stays single-GPU)."""
    from zksnap_circuits_halo2_amd.fields import R_MOD, omega_for

    k, ek = 24, 26
    n, en = 1 << k, 1 << ek

    def timed(fn, reps):
        prewarm(fn, torch, 1, PREWARM_MS / 2)          # (at least one call; the clocks are up when the timed calls start)
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3

    t0, dd = F.fr_encode([24242424])[0], F.fr_encode([0x9E3779B97F4A7C15])[0]
    g = torch.empty(n * 8, dtype=torch.int64, device=dev)
    _lib.check(lib.zkhip_g1_gen_walk_device(t0.ctypes.data, dd.ctypes.data, n, g.data_ptr(), stream))
    torch.cuda.synchronize()
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(g.data_ptr(), n, C.byref(h)))
    del g
    torch.cuda.empty_cache()
    sc = torch.from_numpy(synth_scalars(n, 2424).view(np.int64)).to(dev)
    ext = torch.from_numpy(synth_scalars(en, 2626).view(np.int64)).to(dev)
    res = torch.zeros(16, dtype=torch.int64, device=dev)
    om_i = F.fr_encode([pow(omega_for(k), -1, R_MOD)])[0]
    div = F.fr_encode([pow(n, -1, R_MOD)])[0]
    om_e = F.fr_encode([omega_for(ek)])[0]
    om_ei = F.fr_encode([pow(omega_for(ek), -1, R_MOD)])[0]
    div_e = F.fr_encode([pow(en, -1, R_MOD)])[0]

    def msms():
        for _ in range(18):
            _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, sc.data_ptr(), n, res.data_ptr(), stream))

    def ntts():
        for _ in range(13):
            _lib.check(lib.zkhip_ifft_scaled_device(sc.data_ptr(), om_i.ctypes.data, k, div.ctypes.data, stream))
        for _ in range(13):
            _lib.check(lib.zkhip_ntt_fr_device(ext.data_ptr(), om_e.ctypes.data, ek, stream))
        _lib.check(lib.zkhip_ifft_scaled_device(ext.data_ptr(), om_ei.ctypes.data, ek, div_e.ctypes.data, stream))

    try:
        ms_msm = timed(msms, 1)
        ms_ntt = timed(ntts, 1)
        ms_all = timed(lambda: (msms(), ntts()), 1)
        one_ntt26 = timed(lambda: _lib.check(lib.zkhip_ntt_fr_device(ext.data_ptr(), om_e.ctypes.data, ek, stream)), 2)
        one_intt24 = timed(lambda: _lib.check(lib.zkhip_ifft_scaled_device(sc.data_ptr(), om_i.ctypes.data, k, div.ctypes.data, stream)), 4)
    finally:
        lib.zkhip_release_bases(h)
        del sc, ext
        torch.cuda.empty_cache()
    out = {"workload": "k=24: 18 MSM 2^24 + 13 iNTT 2^24 + 13 NTT 2^26 + 1 iNTT 2^26, device-resident, ONE card",
           "ms": round(ms_all, 2), "msm_part_ms": round(ms_msm, 2), "ntt_part_ms": round(ms_ntt, 2), "ms_per_msm_2^24": round(ms_msm / 18, 3),
           "ms_per_ntt_2^26": round(one_ntt26, 3), "ms_per_intt_2^24": round(one_intt24, 3), "Mpoints_per_s_2^24": round(n / (ms_msm / 18) / 1e3, 1),
           "proofs_per_s_msm_ntt_portion": round(1e3 / ms_all, 3),
           "note": "MSM+NTT portion only (no witness, transcript, quotient); the prepared table of 2^24 points is 13 GiB of the card's 288"}
    shard = c4.get("per_shard_ms")
    if shard:
        # 8 cards: every MSM = one 2^21-point shard per card (measured: per_shard_ms of config4_wrapper_k24_msm, the same kernels on 2^21 points) +
        # the exchange (8 x 96 bytes + fold, ~0.03 ms measured at N = 1 with the gather degenerated to a copy).  Transforms, three ways -- all
        # PROJECTIONS from one card's measurements, none a measurement on 8:
        #   unsharded             every transform on the primary card (rounds 1-4)
        #   spread, copy back     round 5, the `_device` fan-out as built (zkhip_set_ntt_fanout(2)): the 13 + 13 batched transforms cut into 8 shares
        #                         (2 + 2 on the busiest card), a secondary card pulls 0.5 GiB per input and pushes 0.5 / 2 GiB per result over ONE xGMI
        #                         link, priced at an ASSUMED 64 GB/s per direction (not measurable on this one-card box) and not overlapped with its kernels
        #   spread, left in place the same split with the results consumed where they were computed (the bound of SURVEY.md 8(e)'s second split: it needs
        #                         the quotient phase re-cut by rows behind an all-to-all, DESIGN.md section 8 -- not built)
        msm_proj = 18 * (shard + 0.03)
        per_card = -(-13 // 8)                                           # 2 of the 13 inverse and 2 of the 13 extended transforms on the busiest card
        kernels_spread = per_card * one_intt24 + per_card * one_ntt26 + one_ntt26        # + the lone 2^26 inverse transform of the quotient
        gib = float(1 << 30)
        xgmi_ms = (per_card * (0.5 + 0.5) + per_card * (0.5 + 2.0)) * gib / 64e9 * 1e3   # in + out of the iNTTs, in + out of the coset NTTs
        proj_unsharded = msm_proj + ms_ntt
        proj_copy_back = msm_proj + max(kernels_spread, per_card * (one_intt24 + one_ntt26) + xgmi_ms)
        proj_in_place = msm_proj + kernels_spread
        out["projection_8_gpus"] = {"ms": round(proj_unsharded, 2), "proofs_per_s_msm_ntt_portion": round(1e3 / proj_unsharded, 3),
                                    "how": f"18 x (per_shard_ms {shard} + 0.03 exchange) + ntt_part_ms {round(ms_ntt, 2)}: MSM / 8, NTT unsharded -- a PROJECTION from one card's measurements, not a measurement on 8",
                                    "ntt_share_of_projection": round(ms_ntt / proj_unsharded, 3),
                                    "transforms_spread_copy_back": {"ms": round(proj_copy_back, 2), "xgmi_ms_per_secondary_card": round(xgmi_ms, 1),
                                                                    "assumed_xgmi_GBps_per_direction": 64,
                                                                    "how": "13 + 13 batched transforms cut into 8 shares (zkhip_set_ntt_fanout(2)), a secondary card's share = its kernels + its peer copies, not overlapped; PROJECTION with an ASSUMED link rate"},
                                    "transforms_spread_left_in_place": {"ms": round(proj_in_place, 2), "proofs_per_s_msm_ntt_portion": round(1e3 / proj_in_place, 3),
                                                                        "ntt_share_of_projection": round(kernels_spread / proj_in_place, 3),
                                                                        "how": f"the busiest card runs {per_card} of the 13 iNTT 2^24 + {per_card} of the 13 NTT 2^26 + the iNTT 2^26, results consumed in place: the BOUND of the second split, needs the row-sharded quotient phase (not built); PROJECTION"}}
    return out


def prover_phases(lib, _lib, F, torch, dev, stream, timed) -> dict:
    """SURVEY.md 8(f) rows 1-3 on the wrapper shape (halo2-lib BaseConfig k = 22, advice [4], lookup [1,0,0], fixed 1:
    /root/reference/aggregator/benches/wrapper_circuit.rs:61-68; degree 4 -> extended k = 24, 7 permutation columns in 4 sets):
    the fused quotient-numerator pass, the grand products of the permutation / lookup arguments, and the multiopen
    evaluations -- the parts of create_proof besides MSM / NTT that run on device-resident columns."""
    import random

    from zksnap_circuits_halo2_amd import evaluation as E

    k, ek = 22, 24
    n, rows = 1 << k, 1 << ek
    gates = [[E.Fixed(i) * (E.Advice(i, 0) + E.Advice(i, 1) * E.Advice(i, 2) - E.Advice(i, 3))] for i in range(4)]
    cs = E.ConstraintSystem(num_fixed=6, num_advice=5, num_instance=1, gates=gates, lookups=[E.Lookup([E.Advice(4)], [E.Fixed(5)])],
                            permutation_columns=[("advice", i) for i in range(5)] + [("fixed", 4), ("instance", 0)],
                            blinding_factors=5, degree=4)
    qc = E.quotient_columns(cs)
    rng = random.Random(22)
    beta, gamma, theta, y, x = (rng.randrange(F.R_MOD) for _ in range(5))
    cols = []
    for _ in range(qc.total):
        t = torch.randint(0, 1 << 62, (rows, 4), dtype=torch.int64, device=dev)
        t[:, 3] &= (1 << 61) - 1                      # canonical Fr words
        cols.append(t)
    outb = torch.zeros(rows * 4, dtype=torch.int64, device=dev)
    polys = [c.data_ptr() + q * n * 32 for c in cols for q in range(4)]      # 116 distinct 2^22-element polynomials
    res = {}
    # quotient: evaluate_h + divide_by_vanishing_poly
    prog = E.evaluate_h_program(cs, k, ek, beta, gamma, theta, y)
    tinv = torch.from_numpy(F.fr_encode([rng.randrange(1, F.R_MOD) for _ in range(4)]).view(np.int64)).to(dev)
    ptrs = [c.data_ptr() for c in cols]

    def quotient():
        prog.run_device(ptrs, ek, outb.data_ptr(), stream=stream)
        _lib.check(lib.zkhip_mul_periodic_device(outb.data_ptr(), rows, tinv.data_ptr(), 4, stream))

    ms_q = timed(quotient, 3)
    reads = len({(o[1], o[2]) for ins in prog.insns for o in ins[2:5] if o[0] == E.SRC_COLUMN})
    res["quotient_numerator_2^24_rows"] = {"ms": round(ms_q, 3), "program_insns": len(prog.insns), "columns": qc.total,
                                           "column_reads_per_row": reads, "Mrows_per_s": round(rows / ms_q / 1e3, 1),
                                           "hbm_frac_algorithmic": round((qc.total + 1) * 32.0 * rows / (ms_q * 1e-3) / 8e12, 4)}
    # grand products: 4 permutation sets (2 + 2 + 2 + 1 columns) + 1 lookup
    num_b, den_b = outb.data_ptr(), outb.data_ptr() + n * 32
    perm_progs = []
    for s_i, width in enumerate((2, 2, 2, 1)):
        perm_progs.append((E.permutation_numerator_program(width, 2 * s_i, beta, gamma, k), E.permutation_denominator_program(width, beta, gamma), width))
    lk_num, lk_den = E.lookup_product_programs(1, 1, beta, gamma, theta)

    def products():
        for pn, pd, w in perm_progs:
            pn.run_device(polys[:w], k, num_b, stream=stream)
            pd.run_device(polys[:2 * w], k, den_b, stream=stream)
            _lib.check(lib.zkhip_fr_grand_product_device(num_b, den_b, n, num_b, stream))
        lk_num.run_device(polys[4:6], k, num_b, stream=stream)
        lk_den.run_device(polys[6:8], k, den_b, stream=stream)
        _lib.check(lib.zkhip_fr_grand_product_device(num_b, den_b, n, num_b, stream))

    ms_p_sets = timed(products, 3)
    # the same 7 permutation columns in ONE call (zkhip_permutation_products_device: the sets chained on the device) + the lookup's product
    pconsts = [F.fr_encode([v_])[0] for v_ in (beta, gamma, E.DELTA, F.omega_for(k))]
    vptr, sptr = (C.c_void_p * 7)(*polys[:7]), (C.c_void_p * 7)(*polys[7:14])
    z_buf = torch.empty((4, n, 4), dtype=torch.int64, device=dev)
    z_all = z_buf.data_ptr()

    def products_one_call():
        _lib.check(lib.zkhip_permutation_products_device(vptr, sptr, 7, 2, k, n - 6, *[c_.ctypes.data for c_ in pconsts], z_all, stream))
        lk_num.run_device(polys[4:6], k, num_b, stream=stream)
        lk_den.run_device(polys[6:8], k, den_b, stream=stream)
        _lib.check(lib.zkhip_fr_grand_product_device(num_b, den_b, n, num_b, stream))

    ms_p = timed(products_one_call, 3)
    del z_buf
    res["grand_products_4_perm_sets_1_lookup_2^22"] = {"ms": round(ms_p, 3), "ms_one_call_per_set": round(ms_p_sets, 3),
                                                       "how": "zkhip_permutation_products_device (all sets, chained on the device) + the lookup's row programs and grand product"}
    # lookup argument: permute_expression_pair on a range-check shaped pair (lookup_bits = 21: inputs below 2^21, table = the range)
    usable = n - 6
    lk_in = torch.zeros((n, 4), dtype=torch.int64, device=dev)
    lk_in[:, 0] = torch.randint(0, 1 << 21, (n,), dtype=torch.int64, device=dev)
    lk_tab = torch.zeros((n, 4), dtype=torch.int64, device=dev)
    lk_tab[:, 0] = torch.arange(n, dtype=torch.int64, device=dev) % (1 << 21)
    # the kernels take Montgomery words: raw integer words are the Montgomery form of a / R, so one row-program multiply by the
    # constant R = 2^256 mod r encodes the two columns on the device (not timed)
    to_mont = E.RowProgram()
    to_mont.emit(E.OP_MUL, 0, to_mont.column(0), to_mont.constant(pow(2, 256, F.R_MOD)))
    for t in (lk_in, lk_tab):
        to_mont.run_device([t.data_ptr()], k, t.data_ptr(), stream=stream)
    ms_l = timed(lambda: _lib.check(lib.zkhip_lookup_permute_device(lk_in.data_ptr(), lk_tab.data_ptr(), usable, num_b, den_b + n * 32, stream)), 3)
    res["lookup_permute_expression_pair_2^22"] = {"ms": round(ms_l, 3)}
    del lk_in, lk_tab
    # multiopen: 40 evaluations, a 40-polynomial linear combination, 4 divisions by (X - x)
    xw = F.fr_encode([x])[0]
    lin = E.linear_combination_program([rng.randrange(F.R_MOD) for _ in range(40)])
    ev = torch.zeros(40 * 4, dtype=torch.int64, device=dev)
    ev_ptrs = (C.c_void_p * 40)(*polys[:40])

    def multiopen():
        _lib.check(lib.zkhip_fr_eval_polynomial_batch_device(ev_ptrs, 40, n, xw.ctypes.data, ev.data_ptr(), stream))
        lin.run_device(polys[:40], k, num_b, stream=stream)
        for i in range(4):
            _lib.check(lib.zkhip_fr_kate_division_device(polys[i], n, xw.ctypes.data, den_b, stream))

    ms_m = timed(multiopen, 3)
    res["multiopen_40_evals_1_lincomb_4_divisions_2^22"] = {"ms": round(ms_m, 3)}
    res["total_ms"] = round(ms_q + ms_p + ms_l + ms_m, 3)
    res["note"] = "device-side share of create_proof besides MSM/NTT; not included: witness generation, transcript hashing (host side in the reference)"
    del cols, outb
    torch.cuda.empty_cache()
    return res


def small_replays(lib, _lib, F, torch, dev, stream, timed) -> dict:
    """MSM+NTT call mix of the small circuits (BASELINE configs[0] / [2]; column counts from SURVEY.md 8d: voter k=13 with
    the documented proxy A=256, L=8; state-transition k=15 with A=4, L=1, P=3), on the batched entry points: all the
    columns of one phase go through one launch set.  Per proof: MSMs of n = A + 2L + P + L + 1 + 3 + ~2,
    iNTT n the same count, extended NTTs A + 1 + 3L + P, one extended iNTT."""
    from zksnap_circuits_halo2_amd.fields import R_MOD, omega_for

    res = {}
    for name, k, A, L, P in (("voter_k13", 13, 256, 8, 64), ("state_transition_k15", 15, 4, 1, 3)):
        n, ek = 1 << k, k + 2
        n_msm = A + 2 * L + P + L + 1 + 3 + 2
        n_ext = A + 1 + 3 * L + P
        t0, dd = F.fr_encode([4242 + k])[0], F.fr_encode([0x9E3779B97F4A7C15])[0]
        g = torch.empty(n * 8, dtype=torch.int64, device=dev)
        _lib.check(lib.zkhip_g1_gen_walk_device(t0.ctypes.data, dd.ctypes.data, n, g.data_ptr(), stream))
        h = C.c_uint64(0)
        _lib.check(lib.zkhip_prepare_bases_device(g.data_ptr(), n, C.byref(h)))
        cols = torch.from_numpy(synth_scalars(n_msm * n, 77 + k).view(np.int64)).to(dev)
        extc = torch.from_numpy(synth_scalars(n_ext << ek, 78 + k).view(np.int64)).to(dev)
        outs = torch.zeros(n_msm * 12, dtype=torch.int64, device=dev)
        om_i = F.fr_encode([pow(omega_for(k), -1, R_MOD)])[0]
        div = F.fr_encode([pow(n, -1, R_MOD)])[0]
        om_e = F.fr_encode([omega_for(ek)])[0]
        om_ei = F.fr_encode([pow(omega_for(ek), -1, R_MOD)])[0]
        div_e = F.fr_encode([pow(1 << ek, -1, R_MOD)])[0]

        def batched():
            _lib.check(lib.zkhip_ifft_scaled_batch_device(cols.data_ptr(), om_i.ctypes.data, k, div.ctypes.data, n_msm, n, stream))
            _lib.check(lib.zkhip_msm_g1_prepared_batch_device(h, 0, cols.data_ptr(), n, n_msm, n, outs.data_ptr(), stream))
            _lib.check(lib.zkhip_ntt_fr_batch_device(extc.data_ptr(), om_e.ctypes.data, ek, n_ext, 1 << ek, stream))
            _lib.check(lib.zkhip_ifft_scaled_device(extc.data_ptr(), om_ei.ctypes.data, ek, div_e.ctypes.data, stream))

        def one_by_one():
            for b in range(n_msm):
                _lib.check(lib.zkhip_ifft_scaled_device(cols.data_ptr() + b * n * 32, om_i.ctypes.data, k, div.ctypes.data, stream))
                _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, cols.data_ptr() + b * n * 32, n, outs.data_ptr() + b * 96, stream))
            for b in range(n_ext):
                _lib.check(lib.zkhip_ntt_fr_device(extc.data_ptr() + (b << ek) * 32, om_e.ctypes.data, ek, stream))
            _lib.check(lib.zkhip_ifft_scaled_device(extc.data_ptr(), om_ei.ctypes.data, ek, div_e.ctypes.data, stream))

        tb, to = timed(batched, 3), timed(one_by_one, 1)
        res[name] = {"workload": f"k={k}: {n_msm} MSM 2^{k} + {n_msm} iNTT 2^{k} + {n_ext} NTT 2^{ek} + 1 iNTT 2^{ek}, device-resident",
                     "batched_ms": round(tb, 3), "proofs_per_s_msm_ntt_portion": round(1e3 / tb, 1), "one_call_per_column_ms": round(to, 3)}
        lib.zkhip_release_bases(h)
        del g, cols, extc, outs
    return res


def cpu_baseline(log_n, d_scalars, d_bases, d_out, n) -> dict:
    """The reference's algorithm (oracle/cpu_ref.c: restatement of halo2-axiom's best_multiexp / best_fft, since `cargo bench`
    cannot run in this pipeline: /root/reference/README.md:64-73 names the commands being stood in for) on the host cores of the
    GPU box.  `value` is the 2^20 MSM at T = the cores this process may use (scheduler affinity capped by the cgroup CPU quota:
    what rayon's default pool would be sized to); a thread sweep and the NTT / wrapper-shape rows of BASELINE.md section 3 follow."""
    from oracle import cpu_ref as Cr

    from zksnap_circuits_halo2_amd import fields as F

    usable = Cr.usable_cores()
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError:
        quota = None
    log_s = min(log_n, 20)
    m = 1 << log_s
    sc = np.ascontiguousarray(d_scalars.cpu().numpy().view(np.uint64).reshape(-1, 4)[:m])
    bs = np.ascontiguousarray(d_bases.cpu().numpy().view(np.uint64).reshape(-1, 8)[:m])

    def msm_s(s_, b_, t_):
        t = time.perf_counter()
        r = Cr.best_multiexp(s_, b_, t_)
        return time.perf_counter() - t, r

    dt, ref = msm_s(sc, bs, usable)
    agree = None
    if m == n:
        got = d_out.cpu().numpy().view(np.uint64)[:12]
        agree = bool(np.array_equal(Cr.jac_to_affine(np.ascontiguousarray(got)), Cr.jac_to_affine(ref)))
    # thread sweep on the same 2^20 inputs (one thread: a 2^17 prefix, scaled per point)
    sweep = {}
    m1 = min(m, 1 << 17)
    dt1, _ = msm_s(np.ascontiguousarray(sc[:m1]), np.ascontiguousarray(bs[:m1]), 1)
    sweep["1"] = {"Mpoints_per_s": round(m1 / dt1 / 1e6, 5), "sample": f"2^{m1.bit_length() - 1} points, {dt1:.2f} s"}
    for t_ in sorted({8, 16, 32, 64, usable, os.cpu_count() or usable}):
        if t_ == usable:
            sweep[str(t_)] = {"Mpoints_per_s": round(m / dt / 1e6, 4), "sample": f"2^{log_s} points, {dt:.2f} s"}
            continue
        d_, _ = msm_s(sc, bs, t_)
        sweep[str(t_)] = {"Mpoints_per_s": round(m / d_ / 1e6, 4), "sample": f"2^{log_s} points, {d_:.2f} s"}
    # NTT rows (best_fft restatement, T = usable) and the wrapper-shape op mix, extrapolated from one op of each kind
    ntt = {}
    for L in (22, 24):
        a = synth_scalars(1 << L, 4000 + L)
        om = F.fr_encode([F.omega_for(L)])[0]
        t = time.perf_counter()
        Cr.best_fft(a, om, L, usable)
        ntt[f"2^{L}"] = {"s": round(time.perf_counter() - t, 3), "threads": usable}
        del a
    m22 = 1 << 22
    sc22 = synth_scalars(m22, 4100)
    reps = m22 // m
    dt22 = 0.0
    for r in range(reps):        # 2^22-point MSM = best_multiexp over 4 x the 2^20 bases (the chunk-per-thread structure makes its time linear in n at fixed window size; c moves from 12 to 13)
        d_, _ = msm_s(np.ascontiguousarray(sc22[r * m:(r + 1) * m]), bs, usable)
        dt22 += d_
    mix = 18 * dt22 + 13 * ntt["2^22"]["s"] + 14 * ntt["2^24"]["s"]
    # the small circuits' op mixes (BASELINE configs[0] / [2]; op counts as in small_replays), one op of each kind timed (best of 3) and multiplied
    small = {}
    for name, k_, n_msm, n_ext in (("voter_k13", 13, 350, 345), ("state_transition_k15", 15, 16, 11)):
        nn = 1 << k_
        s_ = np.ascontiguousarray(sc[:nn]); b_ = np.ascontiguousarray(bs[:nn])
        t_msm = min(msm_s(s_, b_, usable)[0] for _ in range(3))
        def fft_s(L_):
            a_ = synth_scalars(1 << L_, 4200 + L_)
            om_ = F.fr_encode([F.omega_for(L_)])[0]
            best = 1e9
            for _ in range(3):
                c_ = a_.copy()
                t_ = time.perf_counter()
                Cr.best_fft(c_, om_, L_, usable)
                best = min(best, time.perf_counter() - t_)
            return best
        t_n, t_e = fft_s(k_), fft_s(k_ + 2)
        tot = n_msm * (t_msm + t_n) + (n_ext + 1) * t_e
        small[name] = {"s": round(tot, 3), "proofs_per_s_msm_ntt_portion": round(1.0 / tot, 2), "msm_ms": round(t_msm * 1e3, 3), "ntt_ms": round(t_n * 1e3, 3),
                       "ext_ntt_ms": round(t_e * 1e3, 3), "how": f"{n_msm} x (MSM 2^{k_} + iNTT 2^{k_}) + {n_ext + 1} x NTT 2^{k_ + 2}"}
    return {"value": round(m / dt / 1e6, 4), "unit": "Mpoints/s", "cores": usable, "kind": "port",
            "sample": f"one 2^{log_s}-point MSM (the GPU run's inputs), oracle/cpu_ref.c best_multiexp with {usable} threads, {dt:.2f} s wall",
            "cores_usable": usable, "sched_affinity": affinity, "cgroup_cpu_max": quota, "os_cpu_count": os.cpu_count(),
            "gpu_result_matches": agree, "thread_sweep_msm_2^20": sweep, "ntt_best_fft": ntt,
            "msm_2^22_s": round(dt22, 3), "small_circuit_mixes": small,
            "wrapper_shape_mix_k24": {"s": round(18 * 4 * dt22 + 13 * ntt["2^24"]["s"] + 14 * 4 * (26.0 / 24.0) * ntt["2^24"]["s"], 1),
                                      "how": "EXTRAPOLATED, nothing at 2^24 / 2^26 was run on the CPU: 18 x (2^24 MSM = 4 x the timed 2^22) + 13 x (timed 2^24 best_fft) + "
                                             "14 x (2^26 best_fft = 4 x 26/24 x the timed 2^24: n log n)"},
            "wrapper_shape_mix": {"s": round(mix, 2), "proofs_per_s_msm_ntt_portion": round(1.0 / mix, 4),
                                  "how": "18 x (2^22 MSM, timed as 4 x 2^20) + 13 x (2^22 best_fft) + 14 x (2^24 best_fft), one op of each kind timed and multiplied",
                                  "note": "restatement of the reference's CPU algorithms, not `cargo bench`: no witness generation, no transcript, and the 4x64 field multiply here is plain C (halo2curves uses assembly, roughly 2x faster per multiply)"},
            "note": "a reported baseline, not the target: the roofline fraction above is what describes the kernel"}


if __name__ == "__main__":
    main()
