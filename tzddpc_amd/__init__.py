"""tzddpc_amd -- MI355X-native hot path of Tube-Based Zonotopic Data-Driven Predictive Control.

Drop-in surface of the reference package (``tzddpc/__init__.py:1-15``): ``TZDDPC``, ``Data``,
``SystemZonotopes``, ``Theta``, ``DataDrivenDataset`` plus the set objects the reference imports from
``pyzonotope`` (``Zonotope``, ``MatrixZonotope``, ``Interval``, ``concatenate_zonotope``) and ``cplite``
(use ``from tzddpc_amd import cplite as cp`` where the examples ``import cvxpy as cp``).
"""
from . import cplite
from .gain import compute_A_B, compute_control_gain, compute_theta, is_gain_robust, lqr_gain, spectral_radius
from .objects import (Data, DataDrivenDataset, OptimizationProblem, OptimizationProblemVariables, SystemZonotopes, Theta)
from .tzddpc import TZDDPC, TubeZonotope
from .zonotope import Interval, MatrixZonotope, Zonotope, compute_LTI_matrix_zonotope, concatenate_zonotope

__version__ = "0.1.0"
__all__ = ["TZDDPC", "TubeZonotope", "Data", "DataDrivenDataset", "SystemZonotopes", "Theta", "OptimizationProblem",
           "OptimizationProblemVariables", "Zonotope", "MatrixZonotope", "Interval", "concatenate_zonotope",
           "compute_LTI_matrix_zonotope", "compute_theta", "compute_A_B", "compute_control_gain", "is_gain_robust", "lqr_gain", "spectral_radius", "cplite"]
