"""``OpenGoddard.optimize`` compatibility module (see :mod:`opengoddard_amd.optimize`)."""
from opengoddard_amd.optimize import Condition, Dynamics, Guess, Problem  # noqa: F401

__all__ = ["Problem", "Guess", "Condition", "Dynamics"]
