"""``CommandMixer`` of the reference (/root/reference/src/command_mixer.py:32-82), batched.

``CommandMixer`` keeps the reference's constructor and ``read()`` for one arm: non-blocking reads of
K command ports and a weight port, the ``guard_time`` watchdog, the NaN report, and the weighted sum.
The bookkeeping is host work (it is port polling); the sum itself -- the only arithmetic -- runs on
the GPU through ``vfik_mix`` (bit-exact with the reference's left-to-right double sum).

``BatchedCommandMixer`` is the same for B arms on device arrays ``cmds[K][B][n]``.
"""
import math
import time

import numpy as np


class BatchedCommandMixer:
    def __init__(self, engine, n_channels):
        self.engine = engine
        self.K = int(n_channels)
        esz = engine.io_dtype.itemsize
        self._bytes = engine.batch * engine.n * esz
        self.d_cmds = engine.dev_alloc(self.K * self._bytes)
        self.d_out = engine.dev_alloc(self._bytes)

    def mix(self, cmds, weights):
        """cmds: host array (K, B, n); returns host array (B, n) = sum_k cmds[k] * weights[k]."""
        e = self.engine
        a = np.ascontiguousarray(cmds, dtype=e.io_dtype)
        if a.shape != (self.K, e.batch, e.n):
            raise ValueError("cmds must be (%d, %d, %d)" % (self.K, e.batch, e.n))
        e.h2d(self.d_cmds, a)
        e.mix(self.d_cmds, weights, self.d_out)
        out = np.zeros((e.batch, e.n), dtype=e.io_dtype)
        e.d2h(out, self.d_out)
        return out

    def close(self):
        if self.d_cmds:
            self.engine.dev_free(self.d_cmds)
            self.engine.dev_free(self.d_out)
            self.d_cmds = self.d_out = None


class CommandMixer:
    """Drop-in for ``command_mixer.CommandMixer`` (one arm).  ``engine``: a batch-1 ``Engine`` with
    float64 I/O and n joints whose ``vfik_mix`` does the sum."""

    def __init__(self, ports, weight_port, n, guard_time, weights, engine=None, clock=time.time):
        if engine is None:
            raise ValueError("CommandMixer needs an Engine(batch=1, io_dtype=float64): there is no CPU path")
        if engine.batch != 1 or engine.n != n or engine.io_dtype != np.dtype(np.float64):
            raise ValueError("engine must be batch 1, float64 I/O, %d joints" % n)
        self.clock = clock
        self.nChannels = n
        self.ports = ports
        self.weight_port = weight_port
        if len(ports) != len(weights):  # command_mixer.py:37-41
            print("wrong number of initial weights. Resetting to zeros.")
            self.weights = [0.0] * len(ports)
        else:
            self.weights = weights
        self.guard_time = guard_time
        self.last_command = [[0.0] * n for _ in ports]
        self.last_command_time = [self.clock()] * len(ports)
        self._mixer = BatchedCommandMixer(engine, len(ports))

    def read(self):
        if self.weight_port:  # command_mixer.py:48-53
            bottle = self.weight_port.read(False)
            if bottle:
                for i in range(min(bottle.size(), len(self.ports))):
                    self.weights[i] = bottle.get(i).asDouble()
        for p in range(len(self.ports)):  # command_mixer.py:56-69
            bottle = self.ports[p].read(False)
            if bottle and bottle.size() == self.nChannels:
                self.last_command_time[p] = self.clock()
                self.last_command[p] = [bottle.get(i).asDouble() for i in range(self.nChannels)]
            elif self.clock() - self.last_command_time[p] > self.guard_time:
                self.last_command[p] = [0.0] * self.nChannels
            elif bottle:
                print("wrong length for data bottle")
        for i, cmd in enumerate(self.last_command):  # command_mixer.py:71-75
            for j, v in enumerate(cmd):
                if math.isnan(v):
                    print("nan: %d - %d" % (i, j))
        cmds = np.array(self.last_command, dtype=np.float64).reshape(len(self.ports), 1, self.nChannels)
        return self._mixer.mix(cmds, self.weights)[0].tolist()  # command_mixer.py:78-82
