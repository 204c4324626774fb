"""bench.py's own rank launcher (`python bench.py --gpus N` outside torchrun) and its refusal to report a rank count it did not run.
The metric is "Mpoints/sec at 1/2/4/8 MI355X" (BASELINE.json): `--gpus N` must mean N ranks or no number.  CPU only: the ranks of the
self-test rendezvous over gloo and do no GPU work."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith("{")]


def test_gpus_2_starts_two_ranks_itself():
    r = run(["--gpus", "2", "--launcher-selftest", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1, r.stdout            # ONE line, from rank 0
    rec = lines[0]
    assert rec["launcher_selftest"] is True and rec["n_gpus"] == 2 and rec["ranks_seen"] == 2
    assert rec["launched_by"] == "bench.py" and rec["steps"] == 3 and rec["warmup"] == 1


def test_single_rank_needs_no_launcher():
    r = run(["--gpus", "1", "--launcher-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json_lines(r.stdout)[0]
    assert rec["n_gpus"] == 1 and rec["ranks_seen"] == 1 and rec["launched_by"] == "external torchrun"


def test_more_gpus_than_visible_is_refused_not_downgraded():
    """`--gpus N` with fewer than N devices visible (here: none, or the GPU box's single card) exits non-zero and prints no result line."""
    r = run(["--gpus", "64", "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert "refusing" in r.stderr and "64" in r.stderr
    assert "n_gpus" not in r.stdout


def test_world_size_that_disagrees_with_gpus_is_refused():
    r = run(["--gpus", "2", "--launcher-selftest"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "disagree" in r.stderr and "n_gpus" not in r.stdout
    r = run(["--launcher-selftest"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})      # torchrun with two ranks, --gpus left at 1
    assert r.returncode != 0 and "disagree" in r.stderr and "n_gpus" not in r.stdout


def test_external_torchrun_form_is_accepted():
    """The driver's form for N > 1: torchrun starts the ranks, --gpus N matches WORLD_SIZE."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29731",
                        BENCH, "--gpus", "2", "--launcher-selftest"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json_lines(r.stdout)[0]
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["launched_by"] == "external torchrun"


@pytest.mark.gpu
def test_gpus_2_on_a_one_gpu_box_exits_nonzero():
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("box has two or more GPUs: --gpus 2 is a legitimate run here")
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline"])
    assert r.returncode != 0 and "refusing" in r.stderr and "n_gpus" not in r.stdout


def test_gpu_count_comes_from_sysfs_and_openable_render_nodes(tmp_path, monkeypatch):
    """the launcher parent counts GPUs without HIP: KFD topology nodes with simd_count > 0 whose DRM render node can be opened, capped by
    *_VISIBLE_DEVICES (advisor, round 3: torch.cuda.device_count() in the parent may initialise the runtime)"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    nodes, dri = tmp_path / "nodes", tmp_path / "dri"
    nodes.mkdir(); dri.mkdir()
    for idx, (simd, minor) in enumerate([(0, -1), (1024, 128), (1024, 129), (1024, 130)]):     # a CPU node and three GPUs
        d = nodes / str(idx); d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\ndrm_render_minor {minor}\n")
    (dri / "renderD128").write_text(""); (dri / "renderD129").write_text("")                     # the third card's node is not this container's
    monkeypatch.setenv("ZKHIP_BENCH_KFD_NODES", str(nodes)); monkeypatch.setenv("ZKHIP_BENCH_DRI_DIR", str(dri))
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    assert bench.visible_gpus_without_hip() == 2
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    assert bench.visible_gpus_without_hip() == 1
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("ZKHIP_BENCH_KFD_NODES", str(tmp_path / "missing"))
    assert bench.visible_gpus_without_hip() == 0                                               # no KFD driver: no AMD GPU
