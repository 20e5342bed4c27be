"""Host-side logic that needs no GPU: state_dict layouts, weight extraction order,
sharding arithmetic, fitness settings, loud failure off-GPU."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

imp = lambda n: importlib.import_module("gan-discovery-pso_amd." + n)


def test_state_dict_layouts_match_reference_keys(synth):
    gen = imp("utils.util_dcgan").Generator(10, 1, 64)
    assert list(gen.state_dict().keys()) == list(synth.generator_state(10).keys())
    cnn = imp("pso.util_cnn").get_cnn("ResNet50", 1, [0, 2, 3], 3)
    ref = synth.resnet_state("ResNet50", 1, 3)
    assert set(cnn.state_dict().keys()) == set(ref.keys())
    for k, v in cnn.state_dict().items():
        assert tuple(v.shape) == tuple(ref[k].shape), k
    enc = imp("inverter.utils_ae.util_inverter").Encoder(10, 1)
    assert list(enc.state_dict().keys()) == list(synth.encoder_state(10).keys())
    cnn.load_state_dict(synth.to_torch(ref))  # strict: no missing / unexpected keys
    assert cnn.class_to_idx == {0: 0, 2: 1, 3: 2}
    for name, depth in (("ResNet101", 23), ("ResNet152", 36)):
        m = imp("pso.util_cnn").get_cnn(name, 1, [0, 1], 2)
        assert f"layer3.{depth - 1}.conv3.weight" in m.state_dict()
    with pytest.raises(ValueError):
        imp("pso.util_cnn").get_cnn("NoSuchNet", 1, [0, 1], 2)


def test_change_classifier_head(synth):
    cnn = imp("pso.util_cnn").get_cnn("ResNet50", 1, list(range(8)), 8)
    imp("inverter.utils_ae.util_pso_inverter").change_classifier_n_class(cnn)
    assert tuple(cnn.state_dict()["fc.weight"].shape) == (2, 2048) and tuple(cnn.state_dict()["fc.bias"].shape) == (2,)


def test_tensor_extraction_order(synth):
    eng = imp("engine")
    t, meta = eng.generator_tensors(synth.generator_state(100, 1, 64))
    assert meta == dict(z_dim=100, features_g=64, first_kernel=7) and len(t) == 14
    assert t[0].shape == (100, 128, 7, 7) and t[6].shape == (128, 64, 4, 4) and t[12].shape == (64, 1, 4, 4)
    t, meta = eng.resnet_tensors(synth.resnet_state("ResNet50", 1, 8))
    assert meta == dict(depths=(3, 4, 6, 3), n_class=8)
    assert len(t) == 5 + 16 * 15 + 4 * 5 + 2
    assert t[5].shape == (64, 64, 1, 1) and t[20].shape == (256, 64, 1, 1)  # layer1.0.conv1, its projection shortcut
    assert t[-2].shape == (8, 2048)
    t, meta = eng.encoder_tensors(synth.encoder_state(10))
    assert meta == dict(enc_dim=10, features_e=64) and t[4].shape == (10, 128, 7, 7)


def test_shard_bounds_cover_the_swarm():
    sb = imp("engine").shard_bounds
    for n, w in ((512, 8), (64, 3), (7, 4), (4096, 8), (5, 8)):
        b = [sb(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[r][1] == b[r + 1][0] for r in range(w - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1


def test_fitness_spec_modes():
    eng = imp("engine")
    assert eng.FitnessSpec(mode="optimize_in_training").mode_code() == 0
    assert eng.FitnessSpec(mode="optimize_out_training").mode_code() == 1
    with pytest.raises(ValueError):
        eng.FitnessSpec(mode="nope").mode_code()


def test_rescale_torch_quirk():
    r = imp("utils.util_data").rescale_torch
    x = torch.tensor([[-1.0, 0.0], [1.0, 3.0]])
    assert torch.equal(r(x), (x + 1) / 4)
    assert torch.equal(r(x, min_val=0, max_val=0), (x + 1) / 4)  # an explicit 0 counts as "not given"
    assert torch.equal(r(x, min_val=-3.0, max_val=5.0), (x + 3) / 8)


def test_no_cpu_fallback(synth):
    lib = imp("_lib")
    disc = imp("pso.util_discovery").Discovery(0, synth.generator_state(10), synth.resnet_state("ResNet50", 1, 8),
                                               torch.device("cpu"), iid_classes=list(range(8)))
    with pytest.raises(lib.GdpsoError, match="no CPU fallback"):
        disc.fitness(10, np.zeros(10))


# ----------------------------------------------------------------- convolution launch plans
def _describe(lib, n, hin, cin, cout, k, stride, pad, transposed=0, cus=256):
    import ctypes as C
    buf = (C.c_int * 8192)()
    got = lib.lib().gdpso_conv_plan_describe(n, hin, cin, cout, k, stride, pad, transposed, cus, buf, 8192)
    assert got > 0, lib.lib().gdpso_last_error()
    v = list(buf[:got])
    nphase, tile, splits, hout, wout = v[:5]
    pos, phases = 5, []
    for _ in range(nphase):
        ntaps, off_y, off_x, hm, wm, in_y0, in_x0 = v[pos:pos + 7]
        pos += 7
        taps = [tuple(v[pos + 3 * t:pos + 3 * t + 3]) for t in range(ntaps)]
        pos += 3 * ntaps
        phases.append(dict(off=(off_y, off_x), grid=(hm, wm), origin=(in_y0, in_x0), taps=taps))
    assert pos == got
    return dict(tile=tile, splits=splits, hout=hout, wout=wout, phases=phases)


@pytest.mark.parametrize("hin,k,stride,pad", [(7, 3, 1, 1), (7, 3, 2, 1), (4, 3, 1, 1), (4, 3, 2, 1), (2, 3, 1, 1), (2, 3, 2, 1),
                                             (1, 3, 1, 1), (16, 3, 1, 1), (28, 4, 2, 1), (14, 4, 2, 1), (7, 7, 2, 0), (7, 1, 1, 0),
                                             (7, 1, 2, 0), (9, 7, 1, 3), (5, 5, 1, 2)])
def test_conv_phases_cover_exactly_the_in_range_taps(pkg, hin, k, stride, pad):
    """Border-class phases of a padded convolution (host-side planning, no device): every output
    pixel belongs to exactly one phase, and the (input pixel, kernel tap) pairs that phase gives
    it are exactly the taps that fall inside the input -- nothing from the padding, nothing lost."""
    lib = importlib.import_module("gan-discovery-pso_amd._lib")
    d = _describe(lib, 8, hin, 64, 64, k, stride, pad)
    hout = (hin + 2 * pad - k) // stride + 1
    assert d["hout"] == hout and d["wout"] == hout and d["tile"] in (16, 32, 64, 128) and d["splits"] >= 1
    live = sorted({ky * k + kx for oy in range(hout) for ox in range(hout) for ky in range(k) for kx in range(k)
                   if 0 <= oy * stride - pad + ky < hin and 0 <= ox * stride - pad + kx < hin})
    seen = {}
    padded = 0
    for ph in d["phases"]:
        for my in range(ph["grid"][0]):
            for mx in range(ph["grid"][1]):
                o = (my + ph["off"][0], mx + ph["off"][1])
                assert o not in seen and 0 <= o[0] < hout and 0 <= o[1] < hout
                pairs = set()
                for dy, dx, slot in ph["taps"]:
                    iy, ix = my * stride + ph["origin"][0] + dy, mx * stride + ph["origin"][1] + dx
                    if 0 <= iy < hin and 0 <= ix < hin:
                        pairs.add((iy, ix, live[slot]))
                    else:
                        padded += 1
                seen[o] = pairs
    assert len(seen) == hout * hout
    for (oy, ox), pairs in seen.items():
        want = {(oy * stride - pad + ky, ox * stride - pad + kx, ky * k + kx) for ky in range(k) for kx in range(k)
                if 0 <= oy * stride - pad + ky < hin and 0 <= ox * stride - pad + kx < hin}
        assert pairs == want, (oy, ox)
    if len(d["phases"]) > 1 or pad == 0:
        assert padded == 0  # split into classes (or nothing to split): no tap ever reads padding


@pytest.mark.parametrize("hin", [1, 2, 7, 16])
def test_transposed_conv_phases(pkg, hin):
    """ConvTranspose2d(k4, s2, p1) as sub-pixel phases x border classes: output pixel (oy, ox)
    sums input (iy, ix) with kernel index ky = oy - 2 iy + 1 (and the same along x)."""
    lib = importlib.import_module("gan-discovery-pso_amd._lib")
    d = _describe(lib, 8, hin, 128, 64, 4, 2, 1, transposed=1)
    assert d["hout"] == 2 * hin and len(d["phases"]) <= 16
    k_of = [[1, 3], [2, 0]]
    seen = {}
    for ph in d["phases"]:
        py, px = ph["off"][0] & 1, ph["off"][1] & 1
        for my in range(ph["grid"][0]):
            for mx in range(ph["grid"][1]):
                o = (2 * my + ph["off"][0], 2 * mx + ph["off"][1])
                assert o not in seen
                pairs = set()
                for dy, dx, slot in ph["taps"]:
                    iy, ix = my + ph["origin"][0] + dy, mx + ph["origin"][1] + dx
                    assert 0 <= iy < hin and 0 <= ix < hin  # border classes: never a padding tap
                    pairs.add((iy, ix, k_of[py][slot // 2], k_of[px][slot % 2]))
                seen[o] = pairs
    assert len(seen) == 4 * hin * hin
    for (oy, ox), pairs in seen.items():
        want = {(iy, ix, oy - 2 * iy + 1, ox - 2 * ix + 1) for iy in range(hin) for ix in range(hin)
                if 0 <= oy - 2 * iy + 1 < 4 and 0 <= ox - 2 * ix + 1 < 4}
        assert pairs == want, (oy, ox)


def test_conv_plan_shapes(pkg):
    """Tile / split-K choices at the benchmark's batch (tile codes: 128 = 128x128, 64 = 64x64,
    32 = 32x64 over two wave pairs, 16 = 32x32 over four waves): smaller tiles until there is more
    than one workgroup per CU, split-K only below that."""
    lib = importlib.import_module("gan-discovery-pso_amd._lib")
    assert _describe(lib, 512, 7, 64, 64, 3, 1, 1)["tile"] == 64          # 392 tiles: plain 64x64
    p = _describe(lib, 512, 4, 512, 128, 1, 1, 0)                        # 256 tiles of 64x64 -> 512 of 32x64
    assert p["tile"] == 32 and p["splits"] == 1
    p = _describe(lib, 512, 2, 1024, 256, 1, 1, 0)                       # 256 tiles of 32x64 -> 512 of 32x32
    assert p["tile"] == 16 and p["splits"] == 1
    p = _describe(lib, 512, 1, 2048, 512, 1, 1, 0)                       # 128 tiles of 32x64 -> 256 of 32x32, no slabs
    assert p["tile"] == 16 and p["splits"] == 1
    p = _describe(lib, 128, 1, 2048, 512, 1, 1, 0)                       # 64 tiles of 32x32 -> split-K
    assert p["tile"] == 16 and p["splits"] == 4
    assert _describe(lib, 8192, 7, 256, 256, 1, 1, 0)["tile"] == 128      # deep grid: 128x128
    assert _describe(lib, 1, 7, 64, 64, 3, 1, 1)["splits"] >= 1


def test_chain_tile_choice(pkg):
    """Which layers can be steps of the chained launch: those the planner gives one of the three small tile
    shapes, at any batch size (a share of the batch, an eighth or with two teams per XCD a sixteenth, holds
    ceil(n / shares) images; tiles at its end may be partly empty)."""
    lib = importlib.import_module("gan-discovery-pso_amd._lib")
    f = lib.lib().gdpso_chain_tile_describe
    F32, BF16 = 0, 1
    assert f(512, 2, 1024, 256, 1, 1, 0, 256, 1, F32) == 16      # stage-3 conv1: the planner's 32x32 tiles
    assert f(512, 4, 128, 512, 1, 1, 0, 256, 1, F32) == 64       # stage-2 conv3: 64x64
    assert f(512, 4, 128, 128, 3, 1, 1, 256, 1, F32) == 32       # stage-2 conv2
    assert f(512, 1, 2048, 512, 1, 1, 0, 256, 2, F32) == 16      # two teams: 32 images a share
    assert f(256, 1, 2048, 512, 1, 1, 0, 256, 1, F32) == 16      # planned with split-K alone; a chain never splits
    assert f(500, 2, 1024, 256, 1, 1, 0, 256, 1, F32) == 16      # 63 images a share (the last share 59): partly filled tiles
    assert f(5, 2, 1024, 256, 1, 1, 0, 256, 1, F32) == 16        # one image in each of five shares, three shares empty
    assert f(8192, 7, 256, 256, 1, 1, 0, 256, 1, F32) == 0       # long enough for 128x128 tiles: stands alone
    assert f(512, 2, 1024, 256, 1, 1, 0, 256, 1, BF16) == 16
    assert f(512, 2, 1024, 256, 1, 1, 0, 256, 3, F32) < 0


def test_resident_launch_planner(pkg):
    """What the resident launches do with a batch (host logic only): below 64 images the tiled kernels; one image a
    workgroup up to one image per CU, two beyond; batches of several rounds of workgroups with a short last round
    as a one-image launch of its own; the second stage from 160 images on; the generator's launch in fp32 only where
    nearly every CU gets a workgroup, with 16-bit storage wherever the assessor's runs."""
    import ctypes as C
    L = importlib.import_module("gan-discovery-pso_amd._lib")
    if any(os.environ.get(k) for k in ("GDPSO_STAGE1_RESIDENT", "GDPSO_STAGE2_RESIDENT", "GDPSO_STAGE1_RESIDENT_MIN", "GDPSO_STAGE1_RESIDENT_MAX",
                                       "GDPSO_STAGE2_RESIDENT_MIN", "GDPSO_GEN_RESIDENT", "GDPSO_RESIDENT_TAIL")):
        pytest.skip("the planner's defaults are overridden in this process")
    F32, BF16 = 0, 1

    def plan(n, dtype=F32, cus=256):
        out = [C.c_int(-1) for _ in range(4)]
        L.check(L.lib().gdpso_resident_describe(n, cus, dtype, *[C.byref(o) for o in out]))
        return tuple(o.value for o in out)  # stage images, tail images, second stage, generator images

    assert plan(32) == (0, 0, 0, 0)
    assert plan(64) == (1, 0, 0, 0) and plan(64, BF16) == (1, 0, 1, 1)
    assert plan(1, BF16) == (1, 0, 1, 1) and plan(32, BF16) == (1, 0, 1, 1)  # 16-bit storage: one launch at every size
    assert plan(159)[2] == 0 and plan(160)[2] == 1
    assert plan(256) == (1, 0, 1, 1)            # every CU a workgroup: the generator's launch pays in fp32 too
    assert plan(257) == (2, 0, 1, 0) and plan(384) == (2, 0, 1, 0) and plan(384, BF16) == (2, 0, 1, 2)
    assert plan(512) == (2, 0, 1, 2)
    assert plan(768) == (2, 256, 1, 2)          # one and a half rounds: 512 images two a workgroup, 256 one a workgroup
    assert plan(769) == (2, 0, 1, 0)            # a last round of more than one image per CU stays with two (fp32 generator: tiled)
    assert plan(1024) == (2, 0, 1, 2) and plan(1100) == (2, 76, 1, 0) and plan(1100, BF16) == (2, 76, 1, 2)
    assert plan(16384) == (2, 0, 1, 2)
    assert plan(100, cus=64) == (2, 0, 0, 0) and plan(200, cus=64) == (2, 0, 1, 0)  # 64 CUs: 200 = a round of 128 and 72 more
    assert plan(256, cus=64) == (2, 0, 1, 2)
    assert plan(200, cus=64)[1] == 0 and plan(190, cus=64)[1] == 62
    assert L.lib().gdpso_resident_describe(0, 256, F32, None, None, None, None) != 0
    assert L.lib().gdpso_resident_describe(64, 256, 7, None, None, None, None) != 0


def test_conv_window_larger_than_the_tap_table_is_refused(pkg):
    """A k x k window with more taps than a phase's table holds (64), or offsets beyond int8, is an
    argument error at the C ABI, not a write past the table."""
    import ctypes as C
    L = imp("_lib").lib()
    buf = (C.c_int * 8192)()
    assert L.gdpso_conv_plan_describe(4, 16, 64, 64, 8, 1, 0, 0, 256, buf, 8192) > 0      # 64 taps: the most that fit
    assert L.gdpso_conv_plan_describe(4, 16, 64, 64, 9, 1, 0, 0, 256, buf, 8192) < 0      # 81 taps
    assert b"window too large" in L.gdpso_last_error()
    assert L.gdpso_conv_plan_describe(4, 16, 64, 64, 3, 1, 200, 0, 256, buf, 8192) < 0    # offset -200
    assert L.gdpso_conv_plan_describe(4, 16, 64, 64, 11, 1, 5, 0, 256, buf, 8192) < 0


def test_multiply_shift_division_is_exact():
    """The kernels divide row and tile indices by run-time constants with a multiply and a shift
    (constants made on the host by fast_div_for); the same arithmetic on the host against the
    exact quotient: every divisor up to 4096 and a spread of large ones, dividends at every
    multiple's edges, at random and at the top of the range."""
    f = imp("_lib").lib().gdpso_fast_div_check
    rng = np.random.default_rng(5)
    top = 2 ** 31 - 1
    divisors = list(range(1, 4097)) + [int(x) for x in rng.integers(4097, 2 ** 24, 300)] + [2 ** 24, 2 ** 30, top]
    for d in divisors:
        ns = {0, 1, d - 1, d, d + 1, top, top - 1, (top // d) * d, max(0, (top // d) * d - 1)}
        ns.update(int(x) for x in rng.integers(0, top, 6))
        for k in rng.integers(1, max(2, top // d), 4):
            ns.update((int(k) * d - 1, int(k) * d))
        for n in ns:
            if 0 <= n <= top:
                assert f(n, d) == n // d, (n, d)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2`, started directly as the round-end driver starts it, brings up two
    ranks by itself and relays rank 0's one JSON line (dry run: rendezvous, one exchange through
    engine.TorchComm and the max-over-ranks reduction over gloo, no GPU work)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(GDPSO_BENCH_DRYRUN="1", OMP_NUM_THREADS="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["steps"] == 3
    # the keys a real N > 1 line carries: per-phase times with the collective, the spread over the ranks
    assert set(out["phases"]) >= {"generator_ms", "assessor_ms", "pso_update_and_diagnostic_ms", "collective_ms", "how"}
    assert out["phases"]["collective_ms"] is not None and out["phases"]["collective_ms"] >= 0
    assert out["ranks"]["n_ranks_seen"] == 2
    assert out["ranks"]["ms_per_step_max"] == 2.0 and out["ranks"]["ms_per_step_min"] == 1.0 and out["ms_per_step"] == 2.0
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out


def test_evaluator_cache_key_follows_the_weights(synth):
    """Discovery re-packs its device copy when a model's weights change: the key it caches the packed
    evaluator under moves on load_state_dict (in-place copy: version counters) and on a replaced
    tensor (new storage), and stays put otherwise."""
    Discovery = imp("pso.util_discovery").Discovery
    gen = imp("utils.util_dcgan").Generator(10, 1, 64)
    k0 = Discovery._model_key(gen)
    assert Discovery._model_key(gen) == k0
    gen.load_state_dict(synth.to_torch(synth.generator_state(10, 1, 64, seed=5)))
    k1 = Discovery._model_key(gen)
    assert k1 != k0
    with torch.no_grad():
        next(gen.parameters()).mul_(2.0)
    assert Discovery._model_key(gen) != k1
    sd = synth.generator_state(10, 1, 64, seed=5)
    k2 = Discovery._model_key(sd)
    sd[next(iter(sd))] = np.array(sd[next(iter(sd))])
    assert Discovery._model_key(sd) != k2
