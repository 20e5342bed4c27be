"""Worker for tests/test_sharded_gloo.py: one rank of a 2-process swarm on CPU (gloo).

The sharded loop (engine.run_loop + engine.TorchComm) is host logic: which rank owns which
particles, what is exchanged, when.  Here it is driven with a numpy stand-in for the device
swarm (built on the CPU oracle, test infrastructure only) so that the protocol can be
checked without a GPU: every rank must reproduce the reference trajectory bit for bit.
"""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pso_oracle as oracle  # noqa: E402

eng = importlib.import_module("gan-discovery-pso_amd.engine")
lib = importlib.import_module("gan-discovery-pso_amd._lib")


class NumpySwarm:
    """Same phase methods and attributes as engine.HipSwarm, arithmetic in numpy."""

    def __init__(self, fitness_rows, pos, vel, w, rand, lo, n_global, n_ranks, rank, max_iter, table_positions=False):
        self.n, self.d = pos.shape
        self.first_index, self.n_ranks, self.rank, self.n_global = lo, n_ranks, rank, n_global
        self.fitness_rows, self.rand = fitness_rows, rand
        cd = lib.cand_doubles(self.d)
        self.table_positions = table_positions
        if table_positions:  # this rank's positions behind its record in the exchange table
            assert n_global == self.n * n_ranks
            self.exchange_table = torch.zeros(n_ranks, cd + self.n * self.d, dtype=torch.float64)
            self.pos = self.exchange_table[rank, cd:].view(self.n, self.d)
            self.pos.copy_(torch.from_numpy(pos))
        else:
            self.exchange_table = torch.zeros(n_ranks, cd, dtype=torch.float64)
            self.pos = torch.from_numpy(pos.copy())
        self.cand = self.exchange_table[:, :cd]
        self.vel = vel.copy()
        self.pbest_pos = pos.copy()
        self.pbest_val = np.full(self.n, np.inf, np.float32)
        self.ctrl = np.zeros(lib.CTRL_DOUBLES)
        self.ctrl[lib.CTRL_W], self.ctrl[lib.CTRL_GBEST_VAL] = w, np.inf
        self.gbest_pos = np.zeros(self.d)
        self.g_best_val, self.dummy, self.mean_mse, self.hist = [np.inf], [], [], [pos.copy()]

    def control(self):
        return self.ctrl

    def evaluate(self, i, spec):
        self.fit = self.fitness_rows[i - 1]

    def local_best(self, i):
        better = self.fit < self.pbest_val
        self.pbest_val = np.where(better, self.fit, self.pbest_val)
        self.pbest_pos = np.where(better[:, None], self.pos.numpy(), self.pbest_pos)
        k = int(np.argmin(self.pbest_val))
        self.cand[self.rank, 0] = float(self.pbest_val[k])
        self.cand[self.rank, 1] = float(self.first_index + k)
        self.cand[self.rank, 2:2 + self.d] = torch.from_numpy(self.pbest_pos[k])

    def global_best(self, i, schedule_inertia, early_stopping, tolerance):
        c = self.cand.numpy()
        best, slot = np.float32(self.ctrl[lib.CTRL_GBEST_VAL]), -1
        for r in range(self.n_ranks):
            v = np.float32(c[r, 0])
            if v < best or (slot >= 0 and v == best and c[r, 1] < c[slot, 1]):
                best, slot = v, r
        if slot >= 0:
            if np.isinf(self.g_best_val[-1]):
                self.g_best_val[-1] = best
            else:
                self.g_best_val.append(best)
            self.ctrl[lib.CTRL_GBEST_VAL] = best
            self.gbest_pos = c[slot, 2:2 + self.d].copy()
        self.dummy.append(best)
        if i > 1 and schedule_inertia:
            self.ctrl[lib.CTRL_W] *= 0.99
        self.ctrl[lib.CTRL_ITERS_DONE] = i

    def move(self, i, w_cogn, w_soci, seed=0):
        r = self.rand[i - 1]
        pos = self.pos.numpy()
        vel = self.ctrl[lib.CTRL_W] * self.vel + (w_soci * r[:, 1:2]) * (self.pbest_pos - pos) + (w_cogn * r[:, 0:1]) * (self.gbest_pos[None] - pos)
        self.vel = vel
        self.pos.copy_(torch.from_numpy(pos + vel))  # in place: the positions may live in the exchange table
        self.hist.append(self.pos.numpy().copy())

    def after_exchange(self, i, schedule_inertia, early_stopping, tolerance, w_cogn, w_soci, seed=0, diag_iteration=0):
        if diag_iteration:
            self.distance_finish(diag_iteration)
        self.global_best(i, schedule_inertia, early_stopping, tolerance)
        self.move(i, w_cogn, w_soci, seed)

    def distance_part(self, i, pos_all):
        assert pos_all.shape == (self.n_global, self.d)
        self.cand[self.rank, self.d + 2 + (i & 1)] = oracle.pairwise_distance_part(pos_all.numpy(), self.rank, self.n_ranks)

    def distance_part_table(self, i):
        cd = lib.cand_doubles(self.d)
        self.distance_part(i, self.exchange_table[:, cd:].reshape(self.n_global, self.d))

    def distance_finish(self, i):
        assert len(self.mean_mse) == i - 1  # one diagnostic per iteration, in order
        total = 0.0
        for r in range(self.n_ranks):
            total += float(self.cand[r, self.d + 2 + (i & 1)])
        self.mean_mse.append(total / (0.5 * self.n_global * (self.n_global - 1)))


def main():
    out_dir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = np.load(os.path.join(ROOT, "tests", "golden", "swarm_n64_t50_z10.npz"))
    n, iters = int(g["n"]), 12
    lo, hi = eng.shard_bounds(n, world, rank)
    table = os.environ.get("GDPSO_TEST_TABLE_POSITIONS") == "1"
    sw = NumpySwarm(g["fitness"][:iters, lo:hi], g["pos"][0][lo:hi], g["vel"][0][lo:hi], float(g["w"][0]),
                    g["rand"][:iters, lo:hi], lo, n, world, rank, iters, table_positions=table)
    comm = eng.TorchComm()
    eng.run_loop(sw, iters, None, float(g["w"][1]), float(g["w"][2]), schedule_inertia=False, early_stopping=False,
                 tolerance=1e-5, diagnostics=True, comm=comm)
    hist = comm.gather_rows(np.stack(sw.hist), 1, n)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), hist=hist, dummy=np.array(sw.dummy, np.float32),
             g_best_val=np.array(sw.g_best_val, np.float32), gbest_pos=sw.gbest_pos, mean_mse=np.array(sw.mean_mse),
             )
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
