"""Data containers of the TZDDPC API surface (reference ``tzddpc/objects.py:33-72``).

Same names and fields as the reference so that example scripts construct them unchanged.  The
reference's unused cvxpy-typed ``OptimizationProblem*`` tuples (``objects.py:7-30``) are kept as
plain tuples without the cvxpy types.
"""
from typing import Any, List, NamedTuple

import numpy as np

from .zonotope import Zonotope


class OptimizationProblemVariables(NamedTuple):
    y0: Any
    u: Any
    y: Any
    s_l: Any
    s_u: Any
    beta_u: Any


class OptimizationProblem(NamedTuple):
    variables: OptimizationProblemVariables
    constraints: List[Any]
    objective_function: Any
    problem: Any


class Data(NamedTuple):
    """Input/state data, each of shape T x features (reference ``objects.py:33-40``)."""
    u: np.ndarray
    x: np.ndarray


class DataDrivenDataset(NamedTuple):
    """X+, X-, U- split of the data (reference ``objects.py:43-52``)."""
    Xp: np.ndarray
    Xm: np.ndarray
    Um: np.ndarray
    original_data: Data


class SystemZonotopes(NamedTuple):
    """X0 initial set, U input set, X state set, W process-noise set (reference ``objects.py:55-67``)."""
    X0: Zonotope
    U: Zonotope
    X: Zonotope
    W: Zonotope


class Theta(NamedTuple):
    """Feedback gain and the adversarial model offsets (reference ``objects.py:69-72``)."""
    K: np.ndarray
    deltaA: np.ndarray
    deltaB: np.ndarray
