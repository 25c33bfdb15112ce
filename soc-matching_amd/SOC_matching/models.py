"""Reference-named networks (SOC_matching/models.py:202-393), implemented in `socmx.nets`."""
from socmx.nets import FullyConnectedUNet, SigmoidMLP, TwoBoundarySigmoidMLP  # noqa: F401
