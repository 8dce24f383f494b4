"""pyapes_amd -- MI355X-native FDM stencil + iterative-solve core with the pyapes API.

Drop-in surface (same names and argument meaning as kyoungseoun-chung/pyapes):
``geometry.Box``, ``mesh.Mesh``, ``variables.Field``, the BC factories in
``variables.bcs``, ``solver.fdm.FDM``, ``solver.fdc.FDC``, ``solver.ops.Solver``.
All arithmetic runs in hand-written HIP kernels (``csrc/``) behind the C ABI of
``include/pyapes_hip.h``; PyTorch-ROCm tensors are only the container.  There is
no CPU compute path: operators and solvers raise on a non-GPU mesh.
"""
__version__ = "0.1.0"
