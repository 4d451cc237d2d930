"""CPU: `bench.py --gpus N` / `bench_ppo.py --gpus N` start their own ranks (bench_launch.py): the launcher's
command line, its no-op cases, and a real 2-rank child whose line and exit status are relayed."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return env


@pytest.mark.parametrize("script", ["bench.py", "bench_ppo.py"])
def test_dry_run_prints_the_torchrun_command(script):
    out = subprocess.run([sys.executable, os.path.join(ROOT, script), "--gpus", "8", "--warmup", "2",
                          "--dry-run-launch"], capture_output=True, text=True, env=_clean_env(), timeout=120)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, script))
    assert cmd[i + 1:] == ["--gpus", "8", "--warmup", "2"]        # the script's own arguments, the switch removed


def test_no_launch_for_one_gpu_or_under_a_launcher():
    sys.path.insert(0, ROOT)
    import bench_launch
    assert bench_launch.maybe_self_launch("bench.py", ["--steps", "3"], env={}) is None
    assert bench_launch.maybe_self_launch("bench.py", ["--gpus", "1"], env={}) is None
    assert bench_launch.maybe_self_launch("bench.py", ["--gpus=4"], env={"WORLD_SIZE": "4"}) is None
    assert bench_launch.gpus_arg(["--steps", "5", "--gpus=4"]) == 4


def test_two_rank_child_is_started_and_its_line_and_status_come_back(tmp_path):
    """A stand-in script with the benches' opening lines: outside a launcher it must re-run itself as 2 ranks."""
    script = tmp_path / "fake_bench.py"
    script.write_text(textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, %r)
        if __name__ == "__main__":
            import bench_launch
            rc = bench_launch.maybe_self_launch(__file__)
            if rc is not None:
                raise SystemExit(rc)
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = __import__("torch").ones(1)
        dist.all_reduce(t)
        if dist.get_rank() == 0:
            print(json.dumps({"n_gpus": dist.get_world_size(), "sum": t.item(), "argv": sys.argv[1:]}), flush=True)
        dist.destroy_process_group()
        sys.exit(3 if "--fail" in sys.argv else 0)
    """ % ROOT))
    ok = subprocess.run([sys.executable, str(script), "--gpus", "2"], capture_output=True, text=True,
                        env=_clean_env(), timeout=300)
    assert ok.returncode == 0, ok.stderr
    line = json.loads([l for l in ok.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"n_gpus": 2, "sum": 2.0, "argv": ["--gpus", "2"]}
    bad = subprocess.run([sys.executable, str(script), "--gpus", "2", "--fail"], capture_output=True, text=True,
                         env=_clean_env(), timeout=300)
    assert bad.returncode != 0
