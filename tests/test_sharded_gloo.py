"""The sharded (multi-process) loop on CPU: world_size 2 and 3 over gloo.

Checks the host-side protocol of engine.run_loop / engine.TorchComm -- contiguous particle
shards (uneven for 3 ranks), one all-gather of {value, global index, position} records per
iteration, the diagnostic's position all-gather started behind every move and its per-rank sums
riding in the next records, history collection -- against the
trajectory recorded from the reference.  The device kernels are covered by the GPU tests
(test_gpu_swarm.py::test_logical_shards_match_unsharded runs the same protocol on HIP)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden


@pytest.mark.parametrize("world,table,beside", [(2, False, False), (2, True, False), (3, False, False), (3, False, True)])
def test_sharded_loop_matches_reference(tmp_path, world, table, beside):
    """One all-gather of the candidate records per iteration.  world 2, equal shards: the positions
    either follow every move in a gather of their own (the ranks' distance sums then ride in the next
    records) or lie in the exchange table behind the records (no other collective, the diagnostic two
    exchanges behind); world 3: unequal shards (22 + 21 + 21), positions gathered as padded blocks."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", GDPSO_TEST_TABLE_POSITIONS="1" if table else "0")
    if beside:  # the positions' gather as an asynchronous collective that the loop waits for one iteration later
        env["GDPSO_COMM_ASYNC_BYTES"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + 4 * world + 2 * int(table) + int(beside)), os.path.join(ROOT, "tests", "sharded_worker.py"), str(tmp_path)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    g = load_golden("swarm_n64_t50_z10")
    iters = 12
    for rank in range(world):
        r = np.load(tmp_path / f"rank{rank}.npz")
        np.testing.assert_array_equal(r["hist"], g["pos"][:iters + 1])          # every rank, whole swarm, bit for bit
        np.testing.assert_array_equal(r["dummy"], g["g_best_val_dummy"][:iters])
        np.testing.assert_allclose(r["mean_mse"], g["mean_mse"][:iters], rtol=1e-12)
        n_imp = len(r["g_best_val"])
        np.testing.assert_array_equal(r["g_best_val"], g["g_best_val"][:n_imp])
