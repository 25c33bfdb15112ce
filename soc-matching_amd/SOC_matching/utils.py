"""Reference-named helpers of SOC_matching/utils.py that sit on or next to the hot path."""
import numpy as np
import torch

from socmx.rollout import stochastic_trajectories  # noqa: F401  (utils.py:17-128)


def compute_EMA(value, EMA_value, EMA_coeff=0.01, itr=0):
    """utils.py:389-396: value at itr 0, running mean while itr <= floor(1/coeff), then an EMA."""
    warm = int(np.floor(1 / EMA_coeff))
    if itr == 0:
        return value
    if itr <= warm:
        return (value + itr * EMA_value) / (itr + 1)
    return EMA_coeff * value + (1 - EMA_coeff) * EMA_value
