"""Logical ranks in one process: one thread per rank, each running the sharded loop
(engine.sharded_iterations, or Swarm.optimize with ``comm`` set) on its own swarm, all on one device.

The ranks' kernels go to the device's one default stream in the order the threads enqueue them; a
collective is a set of device-to-device copies between the ranks' own tensors, enqueued between two
barriers: the first makes sure that every rank has enqueued what produces its row before anybody
copies it, the second that every copy is enqueued before a rank's next kernel rewrites its row.
That is the ordering an in-place all-gather gives ranks on different GPUs.

Each rank needs buffers of its own (its own engine.Evaluator: workspace and image buffers); packed
weights may be shared.
"""
import importlib
import threading

import numpy as np
import torch

BARRIER_TIMEOUT_S = 300


class _Shared:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world, timeout=BARRIER_TIMEOUT_S)
        self.rows = [None] * world
        self.objs = [None] * world


class ThreadComm:
    """The methods of engine.TorchComm that the sharded loop and Swarm.optimize use."""
    host_staged = False

    def __init__(self, shared, rank):
        self.s, self.rank, self.world = shared, rank, shared.world
        self.exchanges = 0

    def all_gather(self, out, inp):
        self.s.rows[self.rank] = inp
        self.s.barrier.wait()
        for r in range(self.world):
            if r != self.rank:
                out[r].copy_(self.s.rows[r])
        self.exchanges += 1
        self.s.barrier.wait()

    def start_gather_positions(self, pos, n_global):
        eng = importlib.import_module("gan-discovery-pso_amd.engine")
        self.s.rows[self.rank] = pos
        self.s.barrier.wait()
        allp = torch.cat([self.s.rows[r] for r in range(self.world)])
        self.s.barrier.wait()
        return eng._Gathered(None, allp)

    def gather_rows(self, local, axis, n_global):
        self.s.objs[self.rank] = local
        self.s.barrier.wait()
        out = np.concatenate(list(self.s.objs), axis=axis)
        self.s.barrier.wait()
        return out


def run_ranks(world, fn):
    """fn(rank, comm) on `world` threads; returns the list of results, or re-raises the first failure
    (the other ranks are released from their barriers)."""
    shared = _Shared(world)
    results, errors = [None] * world, [None] * world

    def body(rank):
        try:
            results[rank] = fn(rank, ThreadComm(shared, rank))
        except BaseException as e:  # noqa: BLE001
            errors[rank] = e
            shared.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(BARRIER_TIMEOUT_S + 60)
    real = [e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)]
    if real:
        raise real[0]
    if any(e is not None for e in errors):
        raise errors[[e is not None for e in errors].index(True)]
    assert not any(t.is_alive() for t in threads), "a logical rank did not finish"
    return results
