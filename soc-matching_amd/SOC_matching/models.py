"""Reference-named networks (SOC_matching/models.py:202-393), implemented in `socmx.nets`."""
from socmx.nets import FullyConnectedUNet, SigmoidMLP, TwoBoundarySigmoidMLP  # noqa: F401
from socmx.ground_truth import LinearControl, LowDimControl  # noqa: F401,E402
from socmx.ground_truth import ConstantControl as ConstantControlLinear  # noqa: F401,E402
