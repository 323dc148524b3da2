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


def allowed_cpus():
    """The CPUs this process may run on (its cpuset), sorted."""
    try:
        return sorted(os.sched_getaffinity(0))
    except AttributeError:  # not Linux
        return list(range(os.cpu_count() or 1))


def _sibling_groups(cpus):
    """The allowed logical CPUs grouped by physical core (hardware threads of one core together), cores ordered by their lowest
    CPU number; every CPU a group of its own when the topology is not readable."""
    groups, seen = [], set()
    for c in sorted(cpus):
        if c in seen:
            continue
        sib = {c}
        try:
            txt = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
            for part in txt.split(","):
                lo, _, hi = part.partition("-")
                sib.update(range(int(lo), int(hi or lo) + 1))
        except (OSError, ValueError):
            pass
        sib = sorted(x for x in sib if x in cpus and x not in seen)
        seen.update(sib)
        groups.append(sib)
    return groups


def rank_cpus(local_rank, local_world, cpus=None, max_per_rank=8, groups=None):
    """The CPUs rank `local_rank` of `local_world` pins itself to: a contiguous, disjoint, equal share of the PHYSICAL cores of the
    process's cpuset (`cpus`, default: its affinity mask), with all hardware threads of those cores -- two ranks never share a core --
    at most `max_per_rank` logical CPUs each (the enqueue loop is one thread; HIP's helper threads want a few more).  With fewer cores
    than ranks, several ranks share one core (round-robin).  `groups` (lists of sibling CPUs) overrides the topology read from sysfs."""
    cpus = allowed_cpus() if cpus is None else sorted(cpus)
    if not cpus or local_world < 1 or not (0 <= local_rank < local_world):
        raise ValueError("bad rank %d / world %d / cpus %r" % (local_rank, local_world, cpus))
    cores = [list(g) for g in groups] if groups is not None else _sibling_groups(set(cpus))
    per = len(cores) // local_world
    if per < 1:
        return list(cores[local_rank % len(cores)])
    mine = [c for g in cores[local_rank * per:(local_rank + 1) * per] for c in g]
    if max_per_rank and len(mine) > max_per_rank:   # whole cores first: keep the leading cores' threads
        out = []
        for g in cores[local_rank * per:(local_rank + 1) * per]:
            if len(out) + len(g) > max_per_rank:
                break
            out.extend(g)
        mine = out or mine[:max_per_rank]
    return sorted(mine)


def _parse_cpulist(txt):
    out = set()
    for part in txt.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            out.update(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_local_cpus():
    """Per GPU (amdgpu PCI devices in bus order, which is how the HIP runtime numbers them unless a *_VISIBLE_DEVICES variable
    re-maps them), the CPUs of its NUMA node (`local_cpulist`).  None when that cannot be read or devices are re-mapped."""
    import glob
    if any(os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")):
        return None
    try:
        devs = sorted(d for d in glob.glob("/sys/bus/pci/drivers/amdgpu/*:*:*.*") if os.path.exists(os.path.join(d, "local_cpulist")))
        lists = [_parse_cpulist(open(os.path.join(d, "local_cpulist")).read()) for d in devs]
    except (OSError, ValueError):
        return None
    return lists if lists and all(lists) else None


def rank_cpus_near_gpu(local_rank, local_world, cpus=None, gpu_lists=None, max_per_rank=8, groups=None):
    """rank_cpus, but inside the NUMA node of the rank's GPU: the ranks whose GPUs share a node split THAT node's cores among
    themselves.  Falls back to rank_cpus over the whole cpuset when the node lists are unknown, do not cover every rank, or leave a
    rank without a CPU of the process's cpuset."""
    cpus = allowed_cpus() if cpus is None else sorted(cpus)
    gpu_lists = gpu_local_cpus() if gpu_lists is None else gpu_lists
    if not gpu_lists or len(gpu_lists) < local_world:
        return rank_cpus(local_rank, local_world, cpus, max_per_rank, groups)
    mine = frozenset(gpu_lists[local_rank]) & set(cpus)
    peers = [r for r in range(local_world) if frozenset(gpu_lists[r]) & set(cpus) == mine]
    if not mine:
        return rank_cpus(local_rank, local_world, cpus, max_per_rank, groups)
    sub_groups = None
    if groups is not None:
        sub_groups = [[c for c in g if c in mine] for g in groups]
        sub_groups = [g for g in sub_groups if g]
    return rank_cpus(peers.index(local_rank), len(peers), sorted(mine), max_per_rank, sub_groups)


def pin_rank(local_rank, local_world, log=None, visible_gpus=None):
    """Pin the calling process (a rank, BEFORE its first GPU call: the HIP runtime's threads inherit the mask) to its own
    cores -- near its GPU's NUMA node where sysfs tells -- so that eight 4-us enqueue loops on one host do not migrate over each
    other.  Prints the binding.  VFIK_NO_PIN=1 leaves the process where the launcher put it.  Returns the CPU list, or None when
    nothing was done."""
    if os.environ.get("VFIK_NO_PIN") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    # the GPU's NUMA node is used only when sysfs and the runtime agree on how many GPUs there are (a container that sees one GPU
    # of eight cannot tell which one from its index): visible_gpus = the runtime's device count, from a call that initialises nothing
    lists = gpu_local_cpus()
    if lists is not None and visible_gpus is not None and len(lists) != int(visible_gpus):
        lists = None
    mine = rank_cpus_near_gpu(local_rank, local_world, gpu_lists=lists) if lists else rank_cpus(local_rank, local_world)
    try:
        os.sched_setaffinity(0, mine)
    except OSError as e:  # a cpuset we may not narrow: keep running unpinned
        print("rank %d: sched_setaffinity(%s) failed: %s" % (local_rank, mine, e), file=sys.stderr, flush=True)
        return None
    print("rank %d of %d on this host: pinned to CPUs %s%s" % (local_rank, local_world, ",".join(map(str, mine)),
                                                                " (NUMA node of GPU %d)" % local_rank if lists else ""),
          file=log or sys.stderr, flush=True)
    return mine


def spawn_ranks(argv, world, timeout=None, poll_s=0.05, env=None):
    """Run ``argv`` (a full command line, e.g. [sys.executable, "bench.py", ...]) as ``world`` ranks.

    Returns (returncode, rank0_stdout): returncode is 0 only if every rank exited 0; the first failing
    rank's code otherwise (the others are then terminated, as scripts/vfclik does with its process set),
    or 124 on timeout (every rank terminated).  Rank 0's stdout is drained CONTINUOUSLY by a reader thread
    (a rank that prints more than a pipe holds -- RCCL at NCCL_DEBUG=INFO logs to stdout -- would otherwise
    block in write() and stall its peers in the next collective); the other ranks' stdout is discarded and
    every rank's stderr passes through to ours."""
    import threading
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen(list(argv), env=rank_env(r, world, port, env),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None, text=True))
    chunks = []

    def drain(pipe):
        for line in iter(pipe.readline, ""):
            chunks.append(line)
        pipe.close()

    reader = threading.Thread(target=drain, args=(procs[0].stdout,), daemon=True)
    reader.start()
    deadline = None if timeout is None else time.monotonic() + timeout
    rc = 0
    try:
        pending = set(range(world))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0:  # a crashed peer would leave the others hanging in a collective: end the run
                break
            if deadline is not None and time.monotonic() > deadline:
                rc = 124
                break
            if pending:
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
                    p.wait()
    reader.join(timeout=10)
    return rc, "".join(chunks)


DEFAULT_SPAWN_TIMEOUT_S = 1500.0


def main_spawn(script, args, world, timeout=None):
    """Parent half of ``python script --gpus N`` without an external launcher: re-run the same command
    line as N ranks and relay rank 0's stdout.  Returns the exit code.  A rank stuck in its rendezvous or in RCCL
    initialisation must not hang the caller for ever: after ``timeout`` seconds (default VFIK_SPAWN_TIMEOUT or
    1 500) every rank is terminated and the code is 124 -- fresh processes only, nothing is re-executed."""
    if timeout is None:
        timeout = float(os.environ.get("VFIK_SPAWN_TIMEOUT", DEFAULT_SPAWN_TIMEOUT_S))
    rc, out = spawn_ranks([sys.executable, script] + list(args), world, timeout=timeout if timeout > 0 else None)
    sys.stdout.write(out)
    sys.stdout.flush()
    if rc == 124:
        print("%s: the %d ranks did not finish within %.0f s; terminated" % (os.path.basename(script), world, timeout), file=sys.stderr)
    return rc


# ------------------------------------------------------------------------------------------------------------------
# Which backend carries the barrier and the timing reductions of a multi-rank job (bench.py; the control path has no
# collective).  RCCL's communicator is brought up by a COLLECTIVE call (ncclCommInitRank) that has no timeout of its
# own: a rank that fails before it leaves its peers inside it for good.  So the ranks agree through the rendezvous
# store -- plain key/value traffic, no collective -- BEFORE anyone enters it: every rank publishes a local pre-check
# (RCCL importable, its device usable, which physical device it holds), reads everybody's, and all take the same
# decision from the same data.  Only when every pre-check passed is the communicator tried, under a watchdog that
# ends the process (non-zero) if the attempt does not finish within a bound, so that a rank whose peer died inside
# the collective exits instead of waiting.
# ------------------------------------------------------------------------------------------------------------------
def decide_backend(prechecks):
    """prechecks: one (ok, why, device_id) per rank, in rank order -> ("nccl" | "gloo", reason).  RCCL only when every
    rank's pre-check passed and no two ranks hold the same physical device (RCCL refuses two ranks on one GPU)."""
    for r, (ok, why, _dev) in enumerate(prechecks):
        if not ok:
            return "gloo", "rank %d: %s" % (r, why or "pre-check failed")
    seen = {}
    for r, (_ok, _why, dev) in enumerate(prechecks):
        if dev and dev in seen:
            return "gloo", "ranks %d and %d share device %s" % (seen[dev], r, dev)
        seen[dev] = r
    return "nccl", ""


def agree_on_backend(store, rank, world, precheck, timeout_s=120.0, prefix="pg_pre"):
    """Publish this rank's pre-check, read every rank's, decide.  `store`: a torch.distributed store (set / get; get blocks
    until the key exists or the store's timeout passes -- set here to `timeout_s`, so a rank that never publishes costs its
    peers a bounded wait and an exception, not a hang).  Returns (backend, reason, prechecks)."""
    import datetime
    ok, why, dev = precheck
    store.set_timeout(datetime.timedelta(seconds=timeout_s))
    store.set("%s_%d" % (prefix, rank), "%d|%s|%s" % (1 if ok else 0, (why or "").replace("|", "/")[:160], (dev or "").replace("|", "/")))
    got = []
    for r in range(world):
        o, w, d = store.get("%s_%d" % (prefix, r)).decode().split("|", 2)
        got.append((o == "1", w, d))
    backend, reason = decide_backend(got)
    return backend, reason, got


class Watchdog:
    """Ends the PROCESS (os._exit(code)) if not cancelled within `seconds`: for calls that cannot be interrupted from Python
    (a collective communicator start whose peer is gone).  A fresh daemon thread; the parent launcher sees the exit code and
    terminates the other ranks (spawn_ranks)."""

    def __init__(self, seconds, what, code=3, log=None):
        import threading
        self._t = threading.Timer(seconds, self._fire)
        self._t.daemon = True
        self.what, self.code, self.seconds = what, code, seconds
        self.log = log or (lambda m: print(m, file=sys.stderr, flush=True))

    def _fire(self):
        self.log("[watchdog] %s did not finish within %.0f s: exiting with code %d" % (self.what, self.seconds, self.code))
        os._exit(self.code)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._t.cancel()
        return False
