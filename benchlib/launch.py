"""Process plumbing of the benchmark: the self-launcher of `--gpus N` and the process-group bring-up (never a hang: timeouts + a probe collective)."""
import os
import sys

import torch

BENCH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")


def self_launch(n):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process has touched no GPU API (importing torch
    does not), so it starts N fresh rank processes through `python -m torch.distributed.run` as a CHILD, waits, and exits
    with its code - never an exec from a process that has initialised the GPU."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "8")  # torchrun's default of 1 would throttle nothing here but the noise draws
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH] + sys.argv[1:]
    rc = subprocess.run(cmd, env=env).returncode
    if rc != 0:
        print(f"bench.py: the {n}-rank launch exited with code {rc}", file=sys.stderr)
    sys.exit(rc)


def init_process_group(backend, dev, world, rank, timeout_s):
    """torch.distributed over RCCL (backend "nccl") or gloo, with a timeout on the rendezvous and on every collective (the RCCL watchdog aborts
    the rank when one expires) and a probe all-reduce that builds the communicator HERE, with a message, rather than inside the timed region.
    A failure ends the rank with the library's own error text and exit code 3."""
    import torch.distributed as dist
    from datetime import timedelta

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    try:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=timedelta(seconds=timeout_s))  # RCCL over xGMI
        else:
            dist.init_process_group("gloo", timeout=timedelta(seconds=timeout_s))
        probe = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(probe)
        if int(probe.item()) != world:
            raise RuntimeError(f"all_reduce over {world} ranks returned {probe.item()}")
    except Exception as e:  # noqa: BLE001
        print(f"bench.py rank {rank}: cannot bring up the {backend} process group over {world} ranks: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        sys.exit(3)
    return dist
