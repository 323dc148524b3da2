"""One process per GPU, started from a parent that never touches the GPU.

The reference starts one set of single-threaded processes per arm with ``subprocess.Popen`` and waits
for them (/root/reference/scripts/vfclik:88-105,107-121: terminate the rest when one goes away).  The
batched form is one process per GPU shard; this module is the launcher half of that: it starts
``world`` fresh children with the torch.distributed environment (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT on 127.0.0.1), relays what they print, and reports failure if any child fails.

The parent must not have initialised HIP (no ``torch.cuda.is_available()``, no ``vfik_create``): the
children are fresh interpreters (``subprocess``), never ``fork``/``exec`` of a process that holds a GPU.
"""
import os
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL needs it)
    return env


def spawn_ranks(argv, world, timeout=None, poll_s=0.05, env=None):
    """Run ``argv`` (a full command line, e.g. [sys.executable, "bench.py", ...]) as ``world`` ranks.

    Returns (returncode, rank0_stdout): returncode is 0 only if every rank exited 0; the first failing
    rank's code otherwise (the others are then terminated, as scripts/vfclik does with its process set),
    or 124 on timeout.  Rank 0's stdout is captured and returned; the other ranks' stdout is discarded and
    every rank's stderr passes through to ours."""
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen(list(argv), env=rank_env(r, world, port, env),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None, text=True))
    deadline = None if timeout is None else time.monotonic() + timeout
    rc = 0
    try:
        # rank 0's pipe is drained by communicate() below; poll the others first so that a crashed
        # peer (which would leave rank 0 hanging in a collective) ends the run
        pending = set(range(world))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0:
                break
            if deadline is not None and time.monotonic() > deadline:
                rc = 124
                break
            if pending:
                if 0 in pending:  # keep rank 0's pipe from filling up while we wait
                    try:
                        procs[0].wait(timeout=poll_s)
                    except subprocess.TimeoutExpired:
                        pass
                else:
                    time.sleep(poll_s)
    finally:
        if rc != 0:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
    out = procs[0].stdout.read() if procs[0].stdout else ""
    return rc, out


def main_spawn(script, args, world):
    """Parent half of ``python script --gpus N`` without an external launcher: re-run the same command
    line as N ranks and relay rank 0's stdout.  Returns the exit code."""
    rc, out = spawn_ranks([sys.executable, script] + list(args), world)
    sys.stdout.write(out)
    sys.stdout.flush()
    return rc
