"""Entry point + configuration + setting factory + known-answer checks that need no reference code."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_keys_and_overrides():
    from socmx.config import load_config
    cfg = load_config(["method.setting=double_well", "method.d=10", "optim.M_lr=1e-3", "arch.hdims=[32,16,8]",
                       "method.use_gpu=False"])
    assert cfg.method.setting == "double_well" and cfg.method.d == 10
    assert cfg.optim.M_lr == 1e-3 and cfg.optim.nabla_V_lr == 1e-4 and cfg.optim.adam_eps == 1e-4
    assert cfg.arch.hdims == [32, 16, 8] and cfg.arch.hdims_M == [128, 128]
    assert cfg.method.use_gpu is False and cfg.method.num_steps == 50 and cfg.method.algorithm == "SOCM"
    for key in ("T", "num_steps", "lmbd", "gamma", "gamma2", "gamma3", "d", "use_gpu", "algorithm", "setting", "seed",
                "device", "device_number", "num_iterations", "delta_t_optimal", "delta_x_optimal",
                "scaling_factor_nabla_V", "scaling_factor_M", "compute_control_objective_every", "n_samples_control",
                "use_warm_start", "num_splines", "use_stopping_time"):
        assert key in cfg.method, key


@pytest.mark.parametrize("setting,d", [("OU_quadratic_easy", 2), ("OU_quadratic_hard", 4), ("OU_linear", 6),
                                        ("double_well", 10), ("molecular_dynamics", 1)])
def test_define_variables_surface(setting, d):
    from socmx.config import load_config
    from SOC_matching.experiment_settings.settings import define_variables
    cfg = load_config([f"method.setting={setting}", f"method.d={d}", "method.use_gpu=False", "method.device=cpu",
                       "arch.hdims=[32,16,8]", "arch.hdims_M=[16,16]", "method.num_steps=8"])
    torch.manual_seed(0)
    ts = torch.linspace(0, cfg.method.T, cfg.method.num_steps + 1)
    x0, sigma, optimal_sde, sde, u_warm_start = define_variables(cfg, ts)
    assert x0.shape == (d,) and sigma.shape == (d, d) and u_warm_start is None
    for attr in ("sigma", "lmbd", "dim", "device", "nabla_V", "M", "gamma", "u", "use_learned_control", "T"):
        assert hasattr(sde, attr), attr
    x2, x3 = torch.randn(5, d), torch.randn(3, 5, d)
    t3 = torch.linspace(0, 1, 3)
    for x, t in ((x2, ts[1]), (x3, t3)):
        assert sde.b(t, x).shape == x.shape and sde.nabla_b(t, x).shape == x.shape + (d,)
        assert sde.f(t, x).shape == x.shape[:-1] and sde.nabla_f(t, x).shape == x.shape
        assert sde.g(x).shape == x.shape[:-1] and sde.nabla_g(x).shape == x.shape
        assert sde.control(t, x).shape == x.shape
    assert hasattr(sde, "Phi") == (setting == "molecular_dynamics")
    # nabla_b_T_apply is the contraction the loss uses instead of the dense Jacobian
    v = torch.randn(5, d)
    dense = torch.einsum("bln,bn->bl", sde.nabla_b(ts[1], x2), v)
    np.testing.assert_allclose(sde.problem.nabla_b_T_apply(x2, v).numpy(), dense.numpy(), rtol=1e-5, atol=1e-6)


def test_lq_optimal_control_makes_weights_deterministic():
    """Known answer: under the Riccati control std(lpd+lps+ltw) ~ K^-1/2 and -log E[w] ~ V(x0)."""
    from socmx.config import load_config
    from SOC_matching.experiment_settings.settings import define_variables
    from SOC_matching import utils
    stds = {}
    for K in (50, 200):
        cfg = load_config(["method.setting=OU_quadratic_easy", "method.d=2", "method.use_gpu=False", "method.device=cpu",
                           "arch.hdims=[32,16,8]", "arch.hdims_M=[16,16]", f"method.num_steps={K}"])
        torch.manual_seed(0)
        ts = torch.linspace(0, 1.0, K + 1)
        x0, sigma, optimal_sde, sde, _ = define_variables(cfg, ts)
        r = utils.stochastic_trajectories(optimal_sde, x0.repeat(2048, 1), ts, 1.0)
        lw = r[4] + r[5] + r[6]
        stds[K] = lw.std().item()
        cost = (-(r[4] + r[6])).mean().item()
        assert abs(-torch.log(torch.exp(lw).mean()).item() - cost) < 0.05
    assert stds[50] < 0.09 and stds[200] < 0.045          # 0.060 / 0.030 measured in the survey
    assert 1.5 < stds[50] / stds[200] < 2.6                # ~ sqrt(4)


def test_zero_control_ou_mean_matches_euler_maruyama_closed_form():
    """E[X_K] = (I + A dt)^K x0 exactly for EM with u = 0."""
    from SOC_matching.experiment_settings.OU_quadratic import OU_Quadratic
    from SOC_matching import utils
    torch.manual_seed(1)
    d, K, B = 3, 20, 20000
    A = -0.5 * torch.eye(d) + 0.2 * torch.randn(d, d)
    sde = OU_Quadratic(device="cpu", dim=d, u=lambda t, x: torch.zeros_like(x), A=A, P=torch.eye(d), Q=torch.eye(d),
                       sigma=0.3 * torch.eye(d))
    ts = torch.linspace(0, 1.0, K + 1)
    x0 = torch.tensor([1.0, -0.5, 0.25])
    r = utils.stochastic_trajectories(sde, x0.repeat(B, 1), ts, 1.0)
    Mk = torch.linalg.matrix_power(torch.eye(d) + A * (1.0 / K), K)
    np.testing.assert_allclose(r[0][-1].mean(0).numpy(), (Mk @ x0).numpy(), atol=0.01)


def test_main_runs_a_tiny_training_on_cpu(tmp_path):
    cmd = [sys.executable, os.path.join(ROOT, "soc-matching_amd", "main.py"), "method.setting=double_well", "method.d=3",
           "method.use_gpu=False", "method.num_iterations=4", "method.num_steps=40", "arch.hdims=[16,16,8]",
           "method.delta_t_optimal=0.02", "method.delta_x_optimal=0.02",
           "arch.hdims_M=[8,8]", "method.n_samples_control=32", "+method.n_batches_normalization=2",
           "optim.batch_size=8", "method.gamma=2.0", "method.compute_control_objective_every=2"]
    res = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "Control loss mean" in res.stdout
    run_dir = tmp_path / "outputs" / "runs"
    folders = [p for p in run_dir.iterdir() if p.is_dir()]
    assert len(folders) == 1 and (folders[0] / "last.pkl").exists()     # reference checkpoint convention
    assert folders[0].name.startswith("SOCM_double_well_1.0_1.0_40_False_0_8_")


def test_split_sweep_and_multirun_expansion():
    """Hydra's sweep syntax as the README uses it (README.md:15-60): comma lists outside brackets, -m required."""
    from socmx.config import expand_multirun, split_sweep
    assert split_sweep("SOCM,SOCM_const_M,log-variance") == ["SOCM", "SOCM_const_M", "log-variance"]
    assert split_sweep("[64,64]") == ["[64,64]"] and split_sweep("'a,b',c") == ["'a,b'", "c"]
    jobs, multi = expand_multirun(["method.algorithm=SOCM,rel_entropy", "arch.hdims_M=[64,64]", "method.seed=0,1", "-m"])
    assert multi and len(jobs) == 4
    assert jobs[0] == ["method.algorithm=SOCM", "arch.hdims_M=[64,64]", "method.seed=0"]
    assert jobs[1] == ["method.algorithm=SOCM", "arch.hdims_M=[64,64]", "method.seed=1"]          # the last swept key varies fastest
    assert jobs[3] == ["method.algorithm=rel_entropy", "arch.hdims_M=[64,64]", "method.seed=1"]
    with pytest.raises(ValueError, match="multirun"):
        expand_multirun(["method.algorithm=SOCM,rel_entropy"])
    jobs, multi = expand_multirun(["method.d=3"])
    assert not multi and jobs == [["method.d=3"]]


def test_readme_style_multirun_command_line_runs_every_algorithm(tmp_path):
    """`python main.py method.algorithm='SOCM','log-variance' ... -m` (the form of every README command, README.md:15-60):
    without Hydra the jobs run one after the other in outputs/multiruns/<n>, each with the reference's checkpoint files."""
    cmd = [sys.executable, os.path.join(ROOT, "soc-matching_amd", "main.py"), "method.algorithm=SOCM,log-variance",
           "method.lmbd=1.0", "method.setting=OU_quadratic_easy", "method.gamma=2.0", "method.scaling_factor_M=0.1",
           "optim.M_lr=1e-3", "optim.batch_size=8", "method.num_iterations=3", "method.use_gpu=False", "method.device=cpu",
           "method.d=2", "method.num_steps=10", "arch.hdims=[16,8,8]", "arch.hdims_M=[8,8]", "method.n_samples_control=32",
           "+method.n_batches_normalization=2", "method.compute_control_objective_every=2", "-m"]
    res = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "[multirun] job 0 of 2: method.algorithm=SOCM" in res.stdout
    assert "[multirun] job 1 of 2: method.algorithm=log-variance" in res.stdout
    for num, alg in ((0, "SOCM"), (1, "log-variance")):
        job = tmp_path / "outputs" / "multiruns" / str(num)
        folders = [p for p in job.iterdir() if p.is_dir()]
        assert len(folders) == 1 and folders[0].name.startswith(alg + "_OU_quadratic_easy_"), [f.name for f in folders]
        assert (folders[0] / "last.pkl").exists() and (job / "cmd.sh").exists()
    # on a CPU run the default backend.hip_graph = True falls back to the eager iteration, and says so
    assert "backend.hip_graph: running the eager" in res.stdout


def test_last_pkl_holds_every_series_the_plotting_code_reads(tmp_path):
    """f3: the checkpoint is what the reference's plots.py consumes (plots.py:39-49, 118, 147, 177): unpickle `last.pkl`,
    take `training_info`, and torch.stack every series it reads."""
    import pickle
    cmd = [sys.executable, os.path.join(ROOT, "soc-matching_amd", "main.py"), "method.setting=OU_quadratic_easy", "method.d=2",
           "method.num_steps=10", "method.num_iterations=5", "method.use_gpu=False", "method.device=cpu", "arch.hdims=[16,8,8]",
           "arch.hdims_M=[8,8]", "method.n_samples_control=32", "+method.n_batches_normalization=2", "optim.batch_size=8",
           "method.gamma=2.0", "method.compute_control_objective_every=2"]
    res = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    folders = [p for p in (tmp_path / "outputs" / "runs").iterdir() if p.is_dir()]
    sys.path.insert(0, os.path.join(ROOT, "soc-matching_amd"))
    with open(folders[0] / "last.pkl", "rb") as f:
        solver = pickle.load(f)
    info = solver.training_info
    n_it = 5
    for key in ("time_per_iteration", "EMA_time_per_iteration"):
        assert len(info[key]) == n_it and all(float(v) > 0 for v in info[key])
    for key in ("loss", "EMA_loss", "norm_sqd_diff", "EMA_norm_sqd_diff", "weight_mean", "EMA_weight_mean", "weight_std",
                "EMA_weight_std", "grad_norm_sqd", "EMA_grad_norm_sqd", "sqd_norm_EMA_grad"):
        series = torch.stack([torch.as_tensor(v).reshape(()) for v in info[key]])
        assert series.shape == (n_it,) and torch.isfinite(series).all(), key
    itr = info["control_objective_itr"]
    assert itr == [1, 2, 4, 5]                                    # iteration 0, every 2nd, the last one
    for key in ("control_objective_mean", "control_objective_std_err"):
        series = torch.stack([torch.as_tensor(v).reshape(()) for v in info[key]])
        assert series.shape == (len(itr),) and torch.isfinite(series).all(), key
    assert len(info["trajectories"]) == len(itr) and info["cfg"].method.setting == "OU_quadratic_easy"
    assert solver.num_iterations == n_it and solver.algorithm == "SOCM"


def test_bench_launcher_command_line_and_flag_forwarding():
    """`python bench.py --gpus N` starts its ranks itself: the child command is the contract's launcher line
    (torch.distributed.run, one node, N processes, 127.0.0.1 rendezvous) and carries every flag the ranks must see."""
    sys.path.insert(0, ROOT)
    import bench
    args = bench.make_parser().parse_args(["--gpus", "8", "--steps", "7", "--warmup", "3", "--no-burst", "--no-dist-graph"])
    cmd, env = bench.launch_command(args, 12345, environ={"PATH": os.environ.get("PATH", "")})
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "12345"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    tail = cmd[script + 1:]
    assert tail[:6] == ["--gpus", "8", "--steps", "7", "--warmup", "3"]
    assert "--no-burst" in tail and "--no-dist-graph" in tail
    assert "--no-cpu-baseline" not in tail and "--no-secondary" not in tail and "--force-dist" not in tail
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and int(env["OMP_NUM_THREADS"]) >= 1
    # --spawn (the launcher path on one GPU) makes the child take the sharded code path
    args = bench.make_parser().parse_args(["--gpus", "1", "--spawn", "--defer-graph"])
    cmd, _ = bench.launch_command(args, 1, environ={})
    assert "--force-dist" in cmd and "--defer-graph" in cmd and "--nproc-per-node=1" in cmd
    # the ranks themselves parse what the launcher hands them
    child = bench.make_parser().parse_args(cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:])
    assert child.gpus == 1 and child.force_dist and child.defer_graph


def test_double_well_pde_ground_truth_matches_reference():
    """f2: vectorised 1-D PDE solve + table lookup against the reference's double-loop solver (coarse grid)."""
    from socmx import ground_truth as G
    z = np.load(os.path.join(ROOT, "tests", "golden", "dw_pde_reference.npz"))
    T, delta_t, delta_x, xb = [float(v) for v in z["params"]]
    d = z["kappa"].shape[0]
    tabs = [G.double_well_table_1d(float(z["kappa"][j]), float(z["nu"][j]), 1.0, T, delta_t, delta_x, xb) for j in range(d)]
    ut = np.stack(tabs, axis=2)
    assert ut.shape == z["ut"].shape
    # the reference assembles its tridiagonal matrix from float32 tensor scalars (kappa is fp32), ours is fp64
    # throughout: agreement to the reference's own rounding level (|u| goes up to ~200 at the domain edge)
    np.testing.assert_allclose(ut, z["ut"], rtol=2e-4, atol=2e-3)
    ctrl = G.LowDimControl(torch.from_numpy(z["ut"].copy()), T, xb, d, delta_t, delta_x)   # exact lookup semantics
    ts, xs = torch.from_numpy(z["ts"]), torch.from_numpy(z["xs"])
    np.testing.assert_array_equal(ctrl(ts, xs, t_is_tensor=True).numpy(), z["u_tensor"])
    np.testing.assert_array_equal(ctrl(torch.tensor(float(z["t_scalar"])), xs[3]).numpy(), z["u_scalar"])


def test_double_well_optimal_control_is_a_known_answer():
    """Under the PDE control the importance weights become (nearly) deterministic and
    -log E[w] = control cost = V(x0); both improve with the grid (measured 0.126 / 0.431 vs 0.431 here)."""
    from socmx.config import load_config
    from SOC_matching.experiment_settings.settings import define_variables
    from SOC_matching import utils
    cfg = load_config(["method.setting=double_well", "method.d=3", "method.use_gpu=False", "method.device=cpu",
                       "arch.hdims=[16,16,8]", "arch.hdims_M=[8,8]", "method.num_steps=400",
                       "method.delta_t_optimal=0.002", "method.delta_x_optimal=0.002"])
    torch.manual_seed(0)
    ts = torch.linspace(0, 1.0, 401)
    x0, sigma, optimal_sde, sde, _ = define_variables(cfg, ts)
    r = utils.stochastic_trajectories(optimal_sde, x0.repeat(1024, 1), ts, 1.0)
    lw = r[4] + r[5] + r[6]
    cost = (-(r[4] + r[6])).mean().item()
    assert lw.std().item() < 0.2                         # zero control: 0.48
    assert abs(-torch.log(torch.exp(lw).mean()).item() - cost) < 0.03
