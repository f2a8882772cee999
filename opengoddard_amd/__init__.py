"""MI355X-native pseudospectral NLP assembly/evaluation engine with the OpenGoddard API.

``opengoddard_amd.optimize`` mirrors ``OpenGoddard.optimize`` (Problem / Guess / Condition /
Dynamics); the hot path - LGL construction, collocation defects and the dense
forward-difference Jacobian SLSQP asks for - runs in hand-written HIP kernels for gfx950
(``csrc/``) behind the C ABI in ``include/ogpsx.h``.
"""
__version__ = "0.1.0"
