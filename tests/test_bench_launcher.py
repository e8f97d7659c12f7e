"""bench.py as its own launcher: `python bench.py --gpus N` without WORLD_SIZE must start N ranks itself (the parent
never touches the GPU) and must never silently degrade to one rank.  CPU tests cover the argv / environment
construction and the refusal paths; the GPU test rehearses two ranks on the one GPU of the box over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launcher_command_and_env():
    cmd = bench.launcher_command(4, 29701, ["--gpus", "4", "--steps", "7", "--warmup", "2"], python="/usr/bin/python3",
                                 script="/x/bench.py")
    assert cmd[:3] == ["/usr/bin/python3", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29701"
    i = cmd.index("/x/bench.py")
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]          # the rank sees the caller's flags
    env = bench.launcher_env({"PATH": "/bin", "RANK": "3", "WORLD_SIZE": "9", "LOCAL_RANK": "1", "MASTER_PORT": "1",
                              "SIR_BENCH_SHARE_GPU": "1"})
    assert "RANK" not in env and "WORLD_SIZE" not in env and "LOCAL_RANK" not in env and "MASTER_PORT" not in env
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["SIR_BENCH_SHARE_GPU"] == "1" and env["PATH"] == "/bin"
    assert env["SIR_BENCH_LAUNCHED_BY"] == "bench.py"
    assert bench.launcher_env({"HSA_ENABLE_IPC_MODE_LEGACY": "1"})["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"   # caller's choice wins


def test_pick_json_line():
    lines = ["noise\n", '{"a": 1}\n', '{"metric": "m", "n_gpus": 2}\n', "trailing\n"]
    assert json.loads(bench.pick_json_line(lines))["n_gpus"] == 2
    assert bench.pick_json_line(["x\n", "{broken\n"]) is None


def test_visible_gpu_count_reads_the_kfd_topology_without_hip(tmp_path):
    """CPU nodes (simd_count 0) are not GPUs; *_VISIBLE_DEVICES cuts the count; a missing topology means no GPU."""
    for i, simd in enumerate((0, 0, 1024, 1024, 1024)):
        d = tmp_path / "nodes" / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n")
    nodes = str(tmp_path / "nodes")
    assert bench.visible_gpu_count(nodes, env={}) == 3
    assert bench.visible_gpu_count(nodes, env={"HIP_VISIBLE_DEVICES": "0,2"}) == 2
    assert bench.visible_gpu_count(nodes, env={"ROCR_VISIBLE_DEVICES": "1"}) == 1
    assert bench.visible_gpu_count(str(tmp_path / "absent"), env={}) == 0


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=600)


def test_refuses_instead_of_degrading():
    """More ranks than visible GPUs -> exit code 2 and no result line; WORLD_SIZE != --gpus -> the same."""
    if (bench.visible_gpu_count() or 0) < 2:
        r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"SIR_BENCH_SHARE_GPU": "0"})
        assert r.returncode == 2 and "only" in r.stderr and bench.pick_json_line(r.stdout.splitlines()) is None
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "refusing" in r.stderr and bench.pick_json_line(r.stdout.splitlines()) is None


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_without_an_external_launcher():
    """`SIR_BENCH_SHARE_GPU=1 python bench.py --gpus 2` prints n_gpus 2 and a 2-rank process group (gloo rehearsal of the
    RCCL path: same code, both ranks on cuda:0)."""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "2", "--repeats", "2", "--train-steps", "2", "--no-cpu-baseline"],
             {"SIR_BENCH_SHARE_GPU": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(bench.pick_json_line(r.stdout.splitlines()))
    assert d["n_gpus"] == 2 and d["dist"]["world_size"] == 2 and d["dist"]["backend"] == "gloo"
    assert len(d["dist"]["devices"]) == 2 and d["dist"]["launcher"] == "bench.py"
    assert d["value"] > 0 and d["train"]["value"] > 0 and d["train_aug"]["value"] > 0
    assert d["timed_regions"]["repeats"] == 2 and "roofline" in d["train"]
