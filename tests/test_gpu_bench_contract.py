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


@pytest.mark.parametrize("extra", [["--T", "2048", "--chains", "64", "--no-cpu-baseline", "--no-secondary"],
                                   ["--T", "1024", "--chains", "4", "--no-cpu-baseline", "--no-secondary"],
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
    assert "kernels" in d and all(v["ms_per_step"] >= 0 for v in d["kernels"].values())


def test_bench_full_line_with_general_leg_secondary_legs_and_cpu_baseline():
    """the default structure of the driver's run at reduced sizes: headline + general_path + the C3 / C4 / C5 legs + cpu_baseline"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--T", "4096", "--chains", "64",
                          "--small-secondary"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["model_sharing"].startswith("chain-shared")
    g = d["general_path"]
    assert g["value"] > 0 and g["roofline"]["kernel"].startswith("filter_scan") and g["roofline"]["k3_equivalent_GBps"] > 0
    sec = d["secondary"]
    assert not any("error" in v for v in sec.values()), sec
    assert sec["C3_csmc"]["value"] > 0 and sec["C3_csmc"]["roofline"]["bound"] == "hbm"
    assert sec["C4_lorenz"]["kalman"]["value"] > 0 and sec["C4_lorenz"]["csmc"]["value"] > 0 and sec["C4_lorenz"]["scaling"] == "strong"
    assert all(r["roofline"]["bound"] == "mfma" for r in sec["C5_wide"]["runs"])
    assert sec["C5_wide"]["sweep_16_chains"]["chain_sweeps_per_s"] > 0 and all(b["scalar_filters_per_s"] > 0 for b in sec["C5_wide"]["batched_scalar"])
    assert sec["SV_kalman_general_path"]["order_2"]["value"] > 0 and sec["SV_kalman_general_path"]["order_1_chain_shared"]["value"] > 0   # round 4: the general per-chain path on a real model
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["single_thread_value"] > 0


def test_bench_spawns_its_own_ranks_when_no_launcher_set_rank():
    """`python bench.py --gpus 2` (no RANK in the environment): the parent starts torch.distributed.run as a child before touching the GPU"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--T", "2048", "--chains", "64",
                          "--dist-backend", "gloo", "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0])["n_gpus"] == 2


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
    d = _torchrun(1, ["--T", "2048", "--chains", "64", "--no-cpu-baseline", "--no-secondary"])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["frac"] > 0


@pytest.mark.parametrize("workload", ["kalman", "csmc"])
def test_bench_two_ranks_rehearsal_on_one_gpu(workload):
    """Two ranks sharing this box's GPU (gloo rendezvous; rehearsal mode of bench.py): chains are sharded over ranks, the value is the
    whole-job aggregate, rank 0 alone prints, and the chain-gather returns every rank's chains."""
    extra = ["--dist-backend", "gloo", "--no-cpu-baseline", "--no-secondary"]
    extra += ["--T", "2048", "--chains", "64"] if workload == "kalman" else ["--workload", "csmc", "--T", "512", "--chains", "8", "--N", "128"]
    d = _torchrun(2, extra)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["chains_per_gpu"] == (64 if workload == "kalman" else 8) and "x2" in d["config"]["parallelism"]
