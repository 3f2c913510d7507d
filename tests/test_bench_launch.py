"""bench.py's launch logic: `--gpus N` can never disagree with the ranks that run.
CPU-only: the decision function is pure, and the self-spawn path is exercised with a stub launcher module on PATH-less python."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (importing bench touches neither torch nor the GPU)


def test_single_gpu_plain_python_runs_in_process():
    a = bench.parse_args(["--gpus", "1"])
    assert bench.resolve_launch(a, ["--gpus", "1"], {}) == ("run", 0, 1, 0)


def test_multi_gpu_without_launcher_spawns_torchrun_child():
    argv = ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    a = bench.parse_args(argv)
    kind, cmd = bench.resolve_launch(a, argv, {})
    assert kind == "spawn"
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv  # the child sees the same flags, so its --gpus equals its WORLD_SIZE


def test_under_launcher_world_must_equal_gpus():
    a = bench.parse_args(["--gpus", "8"])
    env = {"RANK": "3", "WORLD_SIZE": "8", "LOCAL_RANK": "3", "MASTER_ADDR": "127.0.0.1"}
    assert bench.resolve_launch(a, [], env) == ("run", 3, 8, 3)
    kind, msg = bench.resolve_launch(a, [], dict(env, WORLD_SIZE="4"))
    assert kind == "error" and "WORLD_SIZE=4" in msg
    a1 = bench.parse_args(["--gpus", "1"])
    kind, msg = bench.resolve_launch(a1, [], env)
    assert kind == "error"


def test_mismatch_exits_nonzero_before_touching_torch():
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 2 and "must agree" in out.stderr and out.stdout.strip() == ""


def test_spawn_passes_the_childs_exit_code_on(tmp_path):
    """`python bench.py --gpus 2` with no RANK: the parent starts `python -m torch.distributed.run ...` as a child and returns its code.
    The child here is a stub package shadowing torch.distributed.run (no GPU in this container), which records its argv."""
    pkg = tmp_path / "torch" / "distributed"
    pkg.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("")
    (pkg / "run.py").write_text("import sys, json, os\nopen(os.environ['STUB_OUT'], 'w').write(json.dumps(sys.argv[1:]))\nsys.exit(7)\n")
    rec = tmp_path / "argv.json"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(PYTHONPATH=str(tmp_path), STUB_OUT=str(rec))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 7, (out.stdout, out.stderr)
    import json
    argv = json.loads(rec.read_text())
    assert "--nproc-per-node=2" in argv and argv[-4:] == ["--gpus", "2", "--steps", "2"]
