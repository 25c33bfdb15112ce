"""Reference-named entry points of SOC_matching/method.py: `NeuralSDE`
(method.py:15-143) and `SOC_Solver` (method.py:146-906), implemented in `socmx`."""
from socmx.sde import NeuralSDE  # noqa: F401
from socmx.solver import SOC_Solver  # noqa: F401
