"""The fused loop under HIP graph capture: gdpso_swarm_iterate only enqueues kernels on the caller's stream (no
allocation, no synchronisation, no host read-back), so the launches of a whole run can be captured into a graph
and replayed; the replay must leave the swarm in the state the direct run leaves it in, bit for bit."""
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,d,iters,dtype", [(64, 10, 6, "f32"), (512, 10, 4, "f32"), (40, 10, 4, "f32"), (512, 100, 3, "bf16"),
                                              (768, 10, 3, "f32")])
def test_captured_loop_replays_to_the_same_state(synth, n, d, iters, dtype):
    """64, 40: the tiled generator (and, at 40, the tiled stem and first stages); 512: the resident launches; 768: a
    short last round of workgroups in launches of their own; bf16 with 100-dimensional latents: the first generator
    layer as a GEMM launch, the PSO update in its multi-workgroup kernels."""
    eng = importlib.import_module("gan-discovery-pso_amd.engine")
    lib = importlib.import_module("gan-discovery-pso_amd._lib")
    dev = torch.device("cuda:0")
    gen = eng.PackedGenerator(synth.generator_state(d, 1, 64, seed=11), dev, dtype=dtype)
    ass = eng.PackedAssessor(synth.resnet_state("ResNet50", 1, 8, seed=12), dev, img_side=28, dtype=dtype)
    ev = eng.Evaluator(gen, ass)
    spec = eng.FitnessSpec(class_index=2)
    sw = eng.HipSwarm(ev, n, d, iters, history=True, diagnostics=True, record_fitness=True)
    r = np.random.RandomState(3).standard_normal((n, 2, d))

    def load():
        sw.load(r[:, 0], (r[:, 1] - 0.5) / 10, 0.73)

    def run():
        sw.iterate(1, iters, spec, 1.496, 1.496, True, False, 1e-5, True, seed=5)

    def state():
        torch.cuda.synchronize()
        res = sw.results()
        return {k: np.array(v, copy=True) for k, v in res.items() if isinstance(v, np.ndarray)}

    load()
    run()
    want = state()
    # the library chains on a device only while it has seen one stream there: capture on a stream of its own
    torch.cuda.synchronize()
    lib.check(lib.lib().gdpso_chain_debug(1, 0, 1))
    try:
        load()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=torch.cuda.Stream(device=dev)):
            run()
        torch.cuda.synchronize()
        for _ in range(2):  # capture does not run anything: the state is still the loaded one before the first replay
            load()
            graph.replay()
            got = state()
            assert set(got) == set(want) and len(want) >= 5
            for k in want:
                np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    finally:
        torch.cuda.synchronize()
        lib.check(lib.lib().gdpso_chain_debug(1, 0, 1))  # forget the capture stream: the tests behind this one chain again
