"""pyapes_amd -- MI355X-native FDM stencil + iterative-solve core with the pyapes API.

Drop-in surface (same names and argument meaning as kyoungseoun-chung/pyapes):
``geometry.Box``, ``mesh.Mesh``, ``variables.Field``, the BC factories in
``variables.bcs``, ``solver.fdm.FDM``, ``solver.fdc.FDC``, ``solver.ops.Solver``.
All arithmetic runs in hand-written HIP kernels (``csrc/``) behind the C ABI of
``include/pyapes_hip.h``; PyTorch-ROCm tensors are only the container.  There is
no CPU compute path: operators and solvers raise on a non-GPU mesh.
"""
__version__ = "0.1.0"


_SUBMODULES = (
    "backend", "geometry", "geometry.basis", "geometry.box", "geometry.cylinder", "mesh", "mesh.mesh",
    "mesh.tools", "variables", "variables.bcs", "variables.fields", "variables.container", "solver",
    "solver.fdm", "solver.fdc", "solver.ops", "solver.linalg", "solver.tools", "solver.types", "solver.rfp",
    "solver.march", "testing", "testing.poisson",
)


def install_as_pyapes(name: str = "pyapes") -> None:
    """Register this package under the reference's import names, so that an unmodified pyapes script
    (``from pyapes.solver.fdm import FDM``; the notebooks' older ``from pyapes.core.solver.fdm import
    FDM`` layout too) runs on the HIP backend after one call::

        import pyapes_amd; pyapes_amd.install_as_pyapes()

    Only the device string changes for the user: ``Mesh(..., device="cuda")``.  Aliases are entries in
    ``sys.modules`` pointing at the very same module objects (no second copy of any class).  Refuses
    to shadow a real ``pyapes`` that is already imported."""
    import importlib
    import sys
    this = sys.modules[__name__]
    have = sys.modules.get(name)
    if have is not None and have is not this:
        raise ImportError(f"install_as_pyapes: a different {name!r} is already imported")
    core = sys.modules.setdefault(name + ".core", this)   # pyapes.core.X is pyapes.X in the notebooks
    sys.modules[name] = this
    for sub in _SUBMODULES:
        mod = importlib.import_module(f"{__name__}.{sub}")
        sys.modules[f"{name}.{sub}"] = mod
        sys.modules[f"{name}.core.{sub}"] = mod
    if not hasattr(this, "core"):
        this.core = core
