"""Self-launch of the multi-GPU benches: `python bench.py --gpus N` with N > 1 and no rank environment starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` as a CHILD process (one
rank per GPU over RCCL), relays its output and returns its exit code.

The parent never touches the GPU (a process that has initialised HIP must not replace itself, and this one does not
exec at all); it is called before anything imports the product package.  Under a launcher (`WORLD_SIZE` set) or
with `--gpus 1` it does nothing.
"""
import os
import socket
import subprocess
import sys


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def gpus_arg(argv):
    """The value of --gpus in argv (both `--gpus N` and `--gpus=N`), default 1."""
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    return n


def launch_argv(script, argv, port=None):
    """Command line of the child: one torch.distributed.run node with --gpus ranks running `script argv`."""
    n = gpus_arg(argv)
    passed = [a for a in argv if a != "--dry-run-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port or _free_port()), script] + passed


def maybe_self_launch(script, argv=None, env=None):
    """-> None when this process should run the bench itself; otherwise the child's exit code (the caller exits
    with it).  `--dry-run-launch` prints the child's command line as one JSON line instead of running it."""
    argv = list(sys.argv[1:] if argv is None else argv)
    env = os.environ if env is None else env
    dry = "--dry-run-launch" in argv
    n = gpus_arg(argv)
    if n <= 1 or "WORLD_SIZE" in env:
        if dry:
            import json
            print(json.dumps({"launch": None, "reason": "single process" if n <= 1 else "already under a launcher"}))
            return 0
        return None
    cmd = launch_argv(os.path.abspath(script), argv)
    if dry:
        import json
        print(json.dumps({"launch": cmd}))
        return 0
    child_env = dict(env)
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on these hosts (RCCL needs it)
    child_env.setdefault("OMP_NUM_THREADS", "4")
    print("[launch] " + " ".join(cmd), file=sys.stderr, flush=True)
    # stdout / stderr are inherited: rank 0's JSON line reaches the caller's stdout as it is printed
    return subprocess.call(cmd, env=child_env)
