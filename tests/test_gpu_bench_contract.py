"""bench.py prints exactly one JSON line with the driver's contract fields (small sizes so it runs in seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"]


@pytest.mark.parametrize("extra", [["--T", "2048", "--chains", "64", "--no-cpu-baseline"],
                                   ["--T", "1024", "--chains", "4", "--no-cpu-baseline"],
                                   ["--workload", "csmc", "--T", "512", "--chains", "8", "--N", "128", "--no-cpu-baseline"]])
def test_bench_json_line(extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"] + extra,
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["value"] > 0 and d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _torchrun(nproc, extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1"] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_under_the_drivers_launcher_one_rank_rccl():
    """The driver's launch line (python -m torch.distributed.run ... bench.py) with one rank: the RCCL group is initialised and the
    barrier / max-over-ranks all-reduce run on it (backend "nccl" = RCCL), so the collective code path is exercised on this one-GPU box."""
    d = _torchrun(1, ["--T", "2048", "--chains", "64", "--no-cpu-baseline"])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["frac"] > 0


@pytest.mark.parametrize("workload", ["kalman", "csmc"])
def test_bench_two_ranks_rehearsal_on_one_gpu(workload):
    """Two ranks sharing this box's GPU (gloo rendezvous; rehearsal mode of bench.py): chains are sharded over ranks, the value is the
    whole-job aggregate, rank 0 alone prints, and the chain-gather returns every rank's chains."""
    extra = ["--dist-backend", "gloo", "--no-cpu-baseline"]
    extra += ["--T", "2048", "--chains", "64"] if workload == "kalman" else ["--workload", "csmc", "--T", "512", "--chains", "8", "--N", "128"]
    d = _torchrun(2, extra)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["chains_per_gpu"] == (64 if workload == "kalman" else 8) and "x2" in d["config"]["parallelism"]
