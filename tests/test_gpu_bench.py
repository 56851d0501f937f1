"""bench.py's multi-rank path on the one-GPU box: `python bench.py --gpus 2` with no launcher in front must start two
fresh rank processes itself (one process per GPU, channels block-partitioned, no data-path collective).
SDRX_BENCH_SHARE_GPU=1 + --backend gloo put both ranks on cuda:0; everything else is the path a driver's
`python bench.py --gpus 8` takes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, timeout=300):
    env = dict(os.environ)
    env["SDRX_BENCH_SHARE_GPU"] = "1"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout)


def test_bench_spawns_two_ranks_on_sharded_cfg4():
    p = _run(["--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--no-cpu", "--log2n", "14",
              "--channels", "65"])  # default workload for N > 1 = cfg4; 65 channels: a ragged partition (32 + 33)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()
    assert p.stdout.decode().strip().splitlines() == lines  # stdout is that line and nothing else (gloo's chatter goes to stderr)
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["steps"] == 3
    assert r["config"]["workload"].startswith("cfg4") and r["config"]["channels_per_gpu"] == 32
    # value = (units of ALL ranks) * steps / max-over-ranks time
    total_units = r["value"] * 1e6 * (r["ms_per_step"] / 1e3)
    assert abs(total_units - 65 * (1 << 14)) < 1.0, total_units
    assert "fir_systolic_kernel<D=10, IQ_INTERLEAVED, EPI_DISC>" == r["roofline"]["kernel"]


def test_bench_single_gpu_point_of_the_same_curve_and_a_failing_rank():
    p = _run(["--gpus", "1", "--workload", "cfg4", "--steps", "2", "--warmup", "1", "--no-cpu", "--log2n", "12",
              "--channels", "16"])
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    r = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert r["n_gpus"] == 1 and r["scaling"] == "strong" and r["config"]["channels_per_gpu"] == 16
    # a rank that dies makes the parent exit non-zero instead of hanging in the other rank's barrier
    p = _run(["--gpus", "2", "--backend", "gloo", "--workload", "cfg3", "--dtype", "f32", "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]


def test_bench_default_backend_keeps_the_control_traffic_on_gloo():
    """No --backend: the process group is 'cpu:gloo,cuda:nccl' and the barrier / scalars travel as CPU tensors, so two
    ranks that share one GPU (where RCCL could not form a communicator) still run: RCCL is never brought up."""
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu", "--log2n", "13", "--channels", "8"])
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    r = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert r["n_gpus"] == 2 and r["config"]["channels_per_gpu"] == 4
    assert r["rccl_ranks"] == 0 and r["rccl"]["rccl_skipped"] == "ranks share a GPU"  # detected, not attempted


def test_bench_rccl_epilogue_on_one_rank():
    """The N > 1 epilogue -- shard.gather_bits over RCCL, rows verified against checksums -- through a one-rank group:
    what a one-GPU box can show of it (the library loads, a communicator comes up, the collective runs on GPU tensors)."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "SDRX_BENCH_SHARE_GPU"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "cfg4", "--steps", "2",
                        "--warmup", "1", "--no-cpu", "--no-extra", "--log2n", "12", "--channels", "65", "--rccl-selftest"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    r = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert r["rccl_ranks"] == 1 and r["rccl"]["gather_verified"] is True, r["rccl"]
    assert r["rccl"]["gather_rows"] == 65 and r["gather_ms"] > 0 and 0.0 < r["rccl"]["gather_ones_fraction"] < 1.0
