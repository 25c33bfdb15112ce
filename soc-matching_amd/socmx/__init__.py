"""socmx -- MI355X-native SOC-matching hot path (rollout + SOCM loss).

Layout:
  _lib        ctypes binding of libsocmx.so (C ABI, include/socmx.h)
  problems    setting descriptors (closed forms as data)
  nets        nabla_V / M networks (reference-compatible parameters)
  sde         NeuralSDE
  rollout     stochastic_trajectories (fused HIP kernel / eager torch)
  loss        SOCM loss (HIP kernels + autograd glue)
  solver      SOC_Solver
  dist        batch sharding over GPUs (RCCL all-reduce of stats + flat gradient)
"""
from . import _lib  # noqa: F401
from .problems import Problem, KIND_OF_SETTING  # noqa: F401
from .sde import NeuralSDE  # noqa: F401
from .rollout import stochastic_trajectories  # noqa: F401
