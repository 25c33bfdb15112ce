"""The SHARDED HIP iteration at world sizes 2 and 3 on ONE GPU (every rank on cuda:0, gloo process group, collectives staged
through host memory: socmx/dist.py transport "staged").  RCCL refuses two ranks on one device, so this is how the code an N-GPU
run executes -- the autograd-free body whose backward kernel writes the control-network gradient straight into the flat
all-reduce buffer, the global 1/((K+1) B) scaling (method.py:717-720), rollouts and noise slices with row0 > 0, uneven splits
(6 / 5 / 5), the second stream's pair-grid-network all-reduce, the stopping-time normaliser, pooled burst statistics -- meets
the kernels before an 8-GPU box does.  References: the reference's own training run (train_*.npz, main.py:280-359), its loss
fixtures, and the same code at world size 1."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from test_host_cpu import GOLDEN

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _launch(world, mode, name, tmp_path, tag, extra_env=None):
    """Start `world` ranks of gpu_dist_child.py and wait for all of them; returns ([json per rank], [npz per rank])."""
    out = str(tmp_path / f"{tag}_{mode}_{world}")
    for attempt in range(2):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "2")
        env.update(extra_env or {})
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "gpu_dist_child.py"), str(r), str(world), port, mode, name, out],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
        outs = []
        for p in procs:
            try:
                o, e = p.communicate(timeout=900)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            outs.append((p.returncode, o, e))
        bringup = any(t in e for _, _, e in outs for t in ("EADDRINUSE", "Address already in use", "Connection refused",
                                                           "DistNetworkError"))
        if all(rc == 0 for rc, _, _ in outs) or not bringup:
            break
    for rc, o, e in outs:
        assert rc == 0, o[-1500:] + e[-3000:]
    js = [json.load(open(out + f".rank{r}.json")) for r in range(world)]
    zs = [np.load(out + f".rank{r}.npz") for r in range(world)]
    return js, zs


def _rel(state, z, pairs):
    num = den = 0.0
    for prefix, tag in pairs:
        for k in [k for k in z.files if k.startswith(prefix)]:
            a = state[tag + k[len(prefix):]]
            num += float(((a - z[k]) ** 2).sum())
            den += float((z[k] ** 2).sum())
    return (num / den) ** 0.5


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", ["train_ou_quadratic_easy_d2", "train_double_well_d10"])
def test_sharded_hip_body_equals_the_reference_training_run(name, world, tmp_path):
    """`Trainer.step` in the DEFAULT schedule of a sharded GPU run (backend.hip_graph True; this transport is not capturable, so
    the autograd-free body runs eagerly) on the reference's own training run: per-iteration loss / mean weight / normaliser
    and the final parameters; 16 rows over 2 ranks (8 / 8) and 3 ranks (6 / 5 / 5); ONE collective per iteration on the main
    stream + one for the pair-grid network's deferred update (from the second iteration on)."""
    js, zs = _launch(world, "train", name, tmp_path, name)
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert [j["rows"] for j in js] == ([[8, 0], [8, 8]] if world == 2 else [[6, 0], [5, 6], [5, 11]])
    n_it = js[0]["iters"]
    for j in js:
        assert j["manual_ok"] and j["hip_graph"] and not j["capture_graphs"], j
        assert j["bodies"] == {"manual": n_it, "eager": 0}, j["bodies"]          # the HIP body, never the autograd iteration
        assert any("sharded run" in m for m in j["logs"]), j["logs"]            # ... and the run said which schedule it took
        # main-stream all-reduce every iteration; the pair-grid network's update of iteration n travels at the start of n + 1,
        # the last one when the trainer is joined
        assert j["collectives_in_steps"] == n_it + (n_it - 1), j
        assert j["collectives_after_join"] == 2 * n_it, j
    r0 = js[0]["rec"]
    np.testing.assert_allclose(r0["loss"], z["train_loss"], rtol=1e-3)
    np.testing.assert_allclose(r0["weight_mean"], z["train_weight_mean"], rtol=1e-3)
    np.testing.assert_allclose(r0["norm"], z["train_norm_const"], rtol=1e-3)
    for j in js[1:]:
        assert j["rec"] == r0                                                     # every rank reports the same reduced values
    assert _rel(zs[0], z, (("final_nablaV.", "V."), ("final_M.", "M."))) < 1e-2
    np.testing.assert_allclose(zs[0]["gamma"], z["final_gamma"], rtol=1e-2)
    for zr in zs[1:]:
        for k in zs[0].files:
            assert np.array_equal(zs[0][k], zr[k]), k                             # replicas stay bit-identical


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_autograd_iteration_on_the_gpu(world, tmp_path):
    """The other sharded schedule (backend.hip_graph False): the eager autograd iteration on the HIP loss kernels with the flat
    gradient all-reduce, same fixture, same bars."""
    name = "train_double_well_d10"
    js, zs = _launch(world, "train_eager", name, tmp_path, name)
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    n_it = js[0]["iters"]
    for j in js:
        assert j["bodies"] == {"manual": 0, "eager": n_it}, j["bodies"]
        assert j["collectives_in_steps"] == n_it, j
    np.testing.assert_allclose(js[0]["rec"]["loss"], z["train_loss"], rtol=1e-3)
    np.testing.assert_allclose(js[0]["rec"]["norm"], z["train_norm_const"], rtol=1e-3)
    assert _rel(zs[0], z, (("final_nablaV.", "V."), ("final_M.", "M."))) < 1e-2
    for zr in zs[1:]:
        for k in zs[0].files:
            assert np.array_equal(zs[0][k], zr[k]), k


@pytest.mark.parametrize("name,world", [("tiny_molecular_dynamics_d2_stopping", 2), ("md_default_d1_K150_B64_stopping", 3)])
def test_stopping_time_training_sharded(name, world, tmp_path):
    """Stopping-time SOCM (molecular_dynamics: per-sample M, the loss normaliser sum(stop_indicators) all-reduced BEFORE the backward,
    method.py:713-715) through `Trainer.step` over a shard: three iterations at world size 2 / 3 (the README run at its default widths:
    22 / 21 / 21 rows) against the same iterations of one process -- losses, weight statistics, normaliser, final parameters."""
    j1, z1 = _launch(1, "train_eager", name, tmp_path, name)
    jn, zn = _launch(world, "train_eager", name, tmp_path, name)
    assert sum(j["rows"][0] for j in jn) == j1[0]["rows"][0]
    for j in jn:
        assert j["bodies"]["eager"] == 3 and j["collectives_in_steps"] == 2 * 3, j       # the normaliser + the flat gradient buffer
    a, b = j1[0]["rec"], jn[0]["rec"]
    for k in ("loss", "weight_mean", "weight_std", "norm"):
        np.testing.assert_allclose(b[k], a[k], rtol=3e-4, err_msg=k)
    num = sum(float(((zn[0][k] - z1[0][k]) ** 2).sum()) for k in z1[0].files)
    den = sum(float((z1[0][k] ** 2).sum()) for k in z1[0].files)
    assert (num / den) ** 0.5 < 5e-5, (num / den) ** 0.5
    for zr in zn[1:]:
        for k in zn[0].files:
            assert np.array_equal(zn[0][k], zr[k]), k


def test_d64_slice_shape_sharded_over_two_ranks(tmp_path):
    """The configs[4] kernels (d = 64: the wide pair-grid-network kernels, the LDS-staged contractions, the general 4-row rollout with a
    dense sigma) under a shard: `cfg5_ou_linear_d64_B256_K3` (the reference's own run at B = 256) split 128 / 128 -- three iterations of
    the sharded HIP body against the same iterations of one process, the first objective against the reference."""
    name = "cfg5_ou_linear_d64_B256_K3"
    j1, z1 = _launch(1, "train", name, tmp_path, name)
    j2, z2 = _launch(2, "train", name, tmp_path, name)
    assert [j["rows"] for j in j2] == [[128, 0], [128, 128]]
    assert all(j["bodies"]["manual"] == 3 and j["bodies"]["eager"] == 0 for j in j2), [j["bodies"] for j in j2]
    a, b = j1[0]["rec"], j2[0]["rec"]
    for k in ("loss", "weight_mean", "weight_std", "norm"):
        np.testing.assert_allclose(b[k], a[k], rtol=3e-4, err_msg=k)
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    np.testing.assert_allclose(b["loss"][0], float(z["loss_objective"]), rtol=5e-4)
    num = sum(float(((z2[0][k] - z1[0][k]) ** 2).sum()) for k in z1[0].files)
    den = sum(float((z1[0][k] ** 2).sum()) for k in z1[0].files)
    assert (num / den) ** 0.5 < 5e-5, (num / den) ** 0.5
    for k in z2[0].files:
        assert np.array_equal(z2[0][k], z2[1][k]), k


def test_headline_configuration_sharded_over_three_ranks(tmp_path):
    """configs[2] at its own size (double_well d=10, K=200, B=128, default widths: the one-row rollout kernel, 43 / 43 / 42 rows)
    through the sharded HIP body: three iterations on the reference's noise against the same iterations of ONE process, and the
    first objective, mean and std of the weights against the reference itself (cfg3_full fixture; normaliser 1.0)."""
    name = "cfg3_full_double_well_d10_K200_B128"
    j1, z1 = _launch(1, "train", name, tmp_path, name)
    j3, z3 = _launch(3, "train", name, tmp_path, name)
    assert [j["rows"] for j in j3] == [[43, 0], [43, 43], [42, 86]]
    assert all(j["bodies"]["manual"] == 3 and j["bodies"]["eager"] == 0 for j in j3)
    a, b = j1[0]["rec"], j3[0]["rec"]
    for k in ("loss", "weight_mean", "weight_std", "norm"):
        np.testing.assert_allclose(b[k], a[k], rtol=2e-4, err_msg=k)
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    np.testing.assert_allclose(b["loss"][0], float(z["loss_objective"]), rtol=5e-4)       # (normaliser 1.0 in the child)
    np.testing.assert_allclose(b["weight_mean"][0], float(z["loss_weight_mean"]), rtol=2e-4)
    np.testing.assert_allclose(b["weight_std"][0], float(z["loss_weight_std"]), rtol=1e-3)
    num = sum(float(((z3[0][k] - z1[0][k]) ** 2).sum()) for k in z1[0].files)
    den = sum(float((z1[0][k] ** 2).sum()) for k in z1[0].files)
    assert (num / den) ** 0.5 < 2e-5, (num / den) ** 0.5
    for zr in z3[1:]:
        for k in z3[0].files:
            assert np.array_equal(z3[0][k], zr[k]), k


@pytest.mark.parametrize("name,world", [("cfg3_double_well_d10_K200", 3), ("tiny_ou_linear_d6", 2), ("ouq20_ou_quadratic_easy_d20_K12", 3)])
def test_philox_rows_do_not_depend_on_the_sharding(name, world, tmp_path):
    """No injected noise: the generator is keyed by the GLOBAL row (include/socmx.h), so rank r's rollout with row0 > 0 is bit
    for bit rows [row0, row0 + B_r) of the one-process launch, and four training iterations on the device-resident key give
    the one-process losses, statistics and parameters (to fp32 summation order across shards)."""
    j1, z1 = _launch(1, "philox", name, tmp_path, name)
    jn, zn = _launch(world, "philox", name, tmp_path, name)
    for j, zr in zip(jn, zn):
        Bl, row0 = j["rows"]
        for k in ("states", "noises"):
            assert np.array_equal(zr[k], z1[0][k][:, row0:row0 + Bl]), (k, j["rows"])
        assert np.array_equal(zr["lpd"], z1[0]["lpd"][row0:row0 + Bl])
        assert j["bodies"]["eager"] == 0 and j["bodies"]["manual"] == 4, j["bodies"]
        assert j["key"][:2] == j1[0]["key"][:2]                                    # every rank advanced its key alike
    assert sum(j["rows"][0] for j in jn) == z1[0]["states"].shape[1]
    np.testing.assert_allclose(jn[0]["rec"], j1[0]["rec"], rtol=2e-4)
    num = sum(float(((zn[0][k] - z1[0][k]) ** 2).sum()) for k in z1[0].files if k[:2] in ("V.", "M."))
    den = sum(float((z1[0][k] ** 2).sum()) for k in z1[0].files if k[:2] in ("V.", "M."))
    assert (num / den) ** 0.5 < 5e-5, (num / den) ** 0.5


def test_sharded_body_with_saved_activations(tmp_path):
    """What every rank of an N-GPU run of the bench does (128 rows each: whole 16-row tiles): the rollout of the shard's rows saves the
    control network's activations, the sharded body's backward runs from them (socmx_unet_backward_saved_f32) and writes into the flat
    all-reduce buffer.  Global batch 64 over two ranks on one device against one process (which saves too) and against one process with the
    re-computing backward: losses, statistics, parameters."""
    name = "cfg3_double_well_d10_K200"
    env = {"SOCMX_TEST_PHILOX_B": "64"}
    j1, z1 = _launch(1, "philox", name, tmp_path, name + "_b64", extra_env=env)
    jn, zn = _launch(2, "philox", name, tmp_path, name + "_b64", extra_env=env)
    assert j1[0]["saved"] and all(j["saved"] for j in jn) and [j["rows"] for j in jn] == [[32, 0], [32, 32]]
    for j, zr in zip(jn, zn):
        Bl, row0 = j["rows"]
        assert np.array_equal(zr["states"], z1[0]["states"][:, row0:row0 + Bl])
        assert j["bodies"]["eager"] == 0 and j["bodies"]["manual"] == 4, j["bodies"]
    np.testing.assert_allclose(jn[0]["rec"], j1[0]["rec"], rtol=2e-4)
    num = sum(float(((zn[0][k] - z1[0][k]) ** 2).sum()) for k in z1[0].files if k[:2] in ("V.", "M."))
    den = sum(float((z1[0][k] ** 2).sum()) for k in z1[0].files if k[:2] in ("V.", "M."))
    assert (num / den) ** 0.5 < 5e-5, (num / den) ** 0.5


@pytest.mark.parametrize("name,world", [("tiny_molecular_dynamics_d2_stopping", 2), ("md_default_d1_K150_B64_stopping", 3),
                                        ("tiny_double_well_d10", 3), ("tiny_ou_linear_d5_B20", 3)])
def test_sharded_loss_call_on_the_gpu_matches_the_reference(name, world, tmp_path):
    """`SOC_Solver.loss` called directly on a shard (HIP kernels): pooled weight statistics, the stopping-time loss's normaliser
    sum(stop_indicators) all-reduced BEFORE the backward (method.py:713-715 -- it scales every gradient), then the flat
    gradient all-reduce: objective, mean / std of the weights and every gradient against the reference's fixture."""
    js, zs = _launch(world, "loss", name, tmp_path, name)
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    stopping = name.endswith("_stopping")
    assert all(j["collectives_in_loss"] == (2 if stopping else 1) for j in js), [j["collectives_in_loss"] for j in js]
    np.testing.assert_allclose(js[0]["objective"], z["loss_objective"], rtol=3e-4)
    np.testing.assert_allclose(js[0]["w_mean"], z["loss_weight_mean"], rtol=2e-4)
    np.testing.assert_allclose(js[0]["w_std"], z["loss_weight_std"], rtol=5e-4)
    from test_host_cpu import build_sde
    sde, _ = build_sde(name)
    names = ["grad_nablaV." + k for k, _ in sde.nabla_V.named_parameters()] + \
            ["grad_M.sigmoid_layers." + k for k, _ in sde.M.sigmoid_layers.named_parameters()] + ["grad_gamma"]
    if stopping:
        names.append("grad_gamma2")
    num = sum(float(((zs[0][f"g{i}"] - z[n]) ** 2).sum()) for i, n in enumerate(names))
    den = sum(float((z[n] ** 2).sum()) for n in names)
    assert (num / den) ** 0.5 < 2e-3, (num / den) ** 0.5
    for zr in zs[1:]:
        for k in zs[0].files:
            assert np.array_equal(zs[0][k], zr[k]), k


def test_control_objective_pools_the_ranks_of_a_shard(tmp_path):
    """method.py:185-221 on a sharded solver: the burst's rows are split over the ranks (each integrates its own slice of the
    injected noise), mean and standard error are pooled with Chan's rule: equal to the one-process call on the same noise."""
    name = "cfg3_double_well_d10_K200"
    j1, _ = _launch(1, "ctrl", name, tmp_path, name)
    j3, _ = _launch(3, "ctrl", name, tmp_path, name)
    for j in j3:
        np.testing.assert_allclose(j["mean"], j1[0]["mean"], rtol=1e-5)
        np.testing.assert_allclose(j["err"], j1[0]["err"], rtol=1e-4)
    assert j1[0]["traj_rows"] == 40 and all(j["traj_rows"] == 40 for j in j3)


def test_bench_line_of_a_two_rank_run_on_one_device(tmp_path):
    """bench.py's N > 1 control flow -- the launcher child under torch.distributed.run, the rank-device gather, K-step blocks between
    barriers with the MAX over ranks, the sharded iteration legs, the secondary configurations over a shard, rank 0's ONE line -- at
    world size 2 on ONE GPU (SOCMX_BENCH_ONE_DEVICE=1: gloo group, collectives staged through the host; the numbers mean nothing,
    the flow is what an N-GPU box will execute with RCCL in its place)."""
    root = os.path.dirname(HERE)
    env = dict(os.environ, SOCMX_BENCH_ONE_DEVICE="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-burst",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{\"metric\"")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and len(line["rank_devices"]) == 2 and line["one_device_debug"]
    assert line["scaling"] == "weak" and line["value"] > 0 and line["config"]["parallelism"].startswith("dp2")
    assert line["shard_transport"] == "staged"
    # the sharded legs: the autograd iteration and the autograd-free body (this transport cannot be captured: no graph leg)
    assert line["socm_ms_per_iter_eager"] > 0 and line["socm_ms_per_iter_eager_body"] > 0 and line["socm_ms_per_iter_graph"] is None
    assert line["socm_ms_per_iter"] == min(line["socm_ms_per_iter_eager"], line["socm_ms_per_iter_eager_body"])
    assert len(line["secondary"]) == 2 and all(e["socm_ms_per_iter_eager_body"] > 0 and np.isfinite(e["last_loss"]) for e in line["secondary"])
    assert "cpu_baseline" not in line                                    # (a rank-0, N = 1 figure)


def test_bench_watchdog_delivers_metric_1_when_the_sharded_legs_do_not_come_back(tmp_path):
    """A multi-GPU bench run whose collectives wedge (first contact of the shard's own RCCL communicators with real peers happens on
    the driver's N-GPU node) must still deliver the scaling measurement: with the watchdog at 1 s -- shorter than bringing the shard
    up and running the first sharded iterations takes -- rank 0 prints the line as far as it got (metric 1 complete, the unfinished
    legs null, the stage named) and the other rank exits non-zero."""
    root = os.path.dirname(HERE)
    env = dict(os.environ, SOCMX_BENCH_ONE_DEVICE="1", SOCMX_BENCH_WATCHDOG_S="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-burst",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    lines = [l for l in res.stdout.splitlines() if l.startswith("{\"metric\"")]
    assert len(lines) == 1, res.stdout[-2000:] + res.stderr[-3000:]
    line = json.loads(lines[0])
    assert "did not finish within 1 s at [" in line["watchdog"]
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["ms_per_step"] > 0 and line["roofline"]["kernel_ms"] > 0
    assert line["socm_ms_per_iter_graph"] is None
    assert res.returncode != 0                       # (the other rank reports the hang as a failure)


def _gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_gpus() < 2, reason="needs one GPU per rank (the boxes this suite has run on so far have one)")
@pytest.mark.parametrize("name", ["train_ou_quadratic_easy_d2", "train_double_well_d10"])
def test_captured_sharded_iteration_over_rccl(name, tmp_path):
    """What a multi-GPU box is for: two ranks on two devices over RCCL -- `Shard()` brings up the package's own communicators,
    `Trainer(hip_graph=True)` captures the iteration with the `ncclAllReduce` launches inside after two eager warm-ups, the ranks
    agree on the capture -- on the reference's own training run.  Skipped wherever there is one GPU."""
    js, zs = _launch(2, "train", name, tmp_path, name, extra_env={"SOCMX_TEST_NCCL": "1"})
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    n_it = js[0]["iters"]
    for j in js:
        assert j["manual_ok"] and j["hip_graph"] and j["capture_graphs"], j
        assert j["captured"] == (1 if n_it > 2 else 0), j                       # two eager warm-ups, then the capture
    np.testing.assert_allclose(js[0]["rec"]["loss"], z["train_loss"], rtol=1e-3)
    np.testing.assert_allclose(js[0]["rec"]["norm"], z["train_norm_const"], rtol=1e-3)
    assert js[1]["rec"] == js[0]["rec"]
    assert _rel(zs[0], z, (("final_nablaV.", "V."), ("final_M.", "M."))) < 1e-2
    for k in zs[0].files:
        assert np.array_equal(zs[0][k], zs[1][k]), k
