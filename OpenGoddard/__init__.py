"""Drop-in import path: ``from OpenGoddard.optimize import Problem, Guess, Condition, Dynamics``
resolves to the MI355X engine's host mirror (:mod:`opengoddard_amd.optimize`)."""
