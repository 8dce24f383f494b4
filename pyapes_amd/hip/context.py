"""``HipContext``: one ``pa_ctx`` per mesh, the only place Python talks to the C ABI.

Tensors are handed over as ``data_ptr()``; the context keeps references to every
tensor whose pointer the library still holds (face-value arrays, tensor
coefficients) so that they outlive the call.
"""
from __future__ import annotations

import ctypes as C
import warnings
from typing import Any, Sequence

import torch
from torch import Tensor

from ..backend import require_gpu
from . import lib as L


def _check(ctx: "HipContext | None", lib: C.CDLL, rc: int, handle: Any = None) -> None:
    if rc == L.PA_OK:
        return
    msg = lib.pa_last_error(handle).decode() if lib is not None else "?"
    raise L.PaError(rc, msg)


class HipContext:
    def __init__(self, mesh: Any):
        if not mesh.is_cuda or not torch.cuda.is_available():
            raise RuntimeError(
                "pyapes_amd: a HIP context needs a mesh on device='cuda' and a visible MI355X; "
                "this backend has no CPU compute path.")
        self.lib = L.load_library()
        self.mesh = mesh
        self.device = mesh.device
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.stream = torch.cuda.current_stream(self.device)
        h = C.c_void_p()
        _check(None, self.lib, self.lib.pa_ctx_create(dev_index, C.c_void_p(self.stream.cuda_stream), C.byref(h)))
        self.h = h
        self.dtype = mesh.dtype.float
        nd = mesh.dim
        n = (C.c_int64 * nd)(*[int(v) for v in mesh.nx])
        dx = (C.c_double * nd)(*mesh.dx_list)
        n0g = int(mesh.global_nx[0])
        self._rc(self.lib.pa_grid_set(self.h, nd, n, dx, L.PA_F64 if self.dtype == torch.float64 else L.PA_F32,
                                      int(mesh.i_off), n0g))
        self._keep: dict[str, Any] = {}
        self._bc_sig: Any = None
        if mesh.coord_sys == "rz":
            self._rc(self.lib.pa_coord_set(self.h, L.PA_COORD_RZ, self._ptr(mesh.x[0].contiguous())))

    def __del__(self):
        try:
            if getattr(self, "h", None) is not None and self.h.value:
                self.lib.pa_ctx_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass

    def set_option(self, name: str, value: bool | int) -> None:
        """Kernel-path switch of this context ("fastpath", "sf", "fold": results do not depend on them;
        "resident": small-mesh solves in one cooperative launch, changes the grouping of the global sums only)."""
        self._rc(self.lib.pa_ctx_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int(0)
        self._rc(self.lib.pa_ctx_get_option(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    def _rc(self, rc: int) -> None:
        _check(self, self.lib, rc, self.h)

    def use_current_stream(self) -> None:
        """Enqueue on torch's CURRENT stream of the mesh device (called at the top of the public entry points:
        tensors the caller just made on that stream are then ordered before the kernels that read them)."""
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream != self.stream.cuda_stream:
            self._rc(self.lib.pa_ctx_set_stream(self.h, C.c_void_p(cur.cuda_stream)))
            self.stream = cur

    def _ptr(self, t: Tensor | None) -> C.c_void_p:
        return C.c_void_p(0 if t is None else t.data_ptr())

    def _field(self, t: Tensor, what: str) -> Tensor:
        """a contiguous scalar field view ``(*nx)`` of the mesh dtype on the GPU"""
        self.use_current_stream()
        require_gpu(t, what)
        if t.dtype != self.dtype:
            raise TypeError(f"pyapes_amd: {what}: tensor dtype {t.dtype} != mesh dtype {self.dtype}")
        if not t.is_contiguous():
            raise ValueError(f"pyapes_amd: {what}: tensor must be contiguous")
        if t.numel() != self.mesh.N:
            raise ValueError(f"pyapes_amd: {what}: expected {self.mesh.N} nodes, got {t.numel()}")
        return t

    # -- boundary conditions -----------------------------------------------------
    def bind_bcs(self, var: Tensor, bcs: Sequence[Any], comp: int = 0, for_rhs: bool = False,
                 types_only: bool = False) -> None:
        """(Re)load the ordered BC list for component ``comp`` of ``var`` ((dim,*nx) tensor).  ``types_only``: the
        faces' types without their values (no callable is evaluated) -- all the stencil rows of ``aop`` depend on."""
        self.use_current_stream()
        self._rc(self.lib.pa_bc_clear(self.h))
        keep = []
        for pos, bc in enumerate(bcs):
            scalar, arr = (0.0, None) if types_only else bc.resolve(var, comp, for_rhs=for_rhs)
            if arr is not None:
                keep.append(arr)
            dxf = self.mesh.face_dxf(bc.bc_face) if bc.bc_type == "neumann" else 0.0
            self._rc(self.lib.pa_bc_set(self.h, self.mesh.face_index(bc.bc_face), pos, L.BC_CODE[bc.bc_type],
                                        float(scalar), self._ptr(arr), float(dxf)))
        self._keep["bc"] = keep

    def apply_bcs(self, var: Tensor, bcs: Sequence[Any], comps: Sequence[int] | None = None) -> None:
        """``linalg._apply_bc_otf``: every component, every face in list order, in place."""
        require_gpu(var, "BC fill")
        for d in (range(var.shape[0]) if comps is None else comps):
            self.bind_bcs(var, bcs, d)
            self._rc(self.lib.pa_apply_bc(self.h, self._ptr(self._field(var[d], "BC fill"))))

    def apply_bc_bound(self, x: Tensor) -> None:
        """BC fill with the list already bound by ``bind_bcs`` (one scalar field)."""
        self._rc(self.lib.pa_apply_bc(self.h, self._ptr(self._field(x, "BC fill"))))

    # -- equation -------------------------------------------------------------------
    def set_terms(self, terms: Sequence[dict]) -> None:
        """terms: dicts with kind, sign, coeff (None|float|Tensor), u (float|Tensor)."""
        self.use_current_stream()
        arr = (L.PaTerm * len(terms))()
        keep = []
        for q, t in enumerate(terms):
            arr[q].kind = t["kind"]
            arr[q].sign = float(t.get("sign", 1.0))
            coeff = t.get("coeff")
            arr[q].has_coeff = 0 if coeff is None else 1
            arr[q].coeff = 0.0
            arr[q].coeff_field = None
            if isinstance(coeff, Tensor):
                cf = torch.broadcast_to(coeff.to(self.device, self.dtype), (1, *self.mesh.nx)).contiguous()
                keep.append(cf)
                arr[q].coeff_field = cf.data_ptr()
            elif coeff is not None:
                arr[q].coeff = float(coeff)
            u = t.get("u", 0.0)
            arr[q].u = 0.0
            arr[q].u_field = None
            if isinstance(u, Tensor):
                uf = self._field(u if u.dim() == self.mesh.dim else u[0], "advection tensor")
                keep.append(uf)
                arr[q].u_field = uf.data_ptr()
            else:
                arr[q].u = float(u)
        self._keep["terms"] = (arr, keep)
        self._rc(self.lib.pa_eq_set(self.h, len(terms), arr))

    def aop(self, x: Tensor, interior_only: bool = False, out: Tensor | None = None) -> Tensor:
        x = self._field(x, "Aop")
        y = torch.empty_like(x) if out is None else self._field(out, "Aop out")
        self._rc(self.lib.pa_aop(self.h, self._ptr(x), self._ptr(y), 1 if interior_only else 0))
        return y

    def rhs_adjust(self, rhs: Tensor) -> None:
        self._rc(self.lib.pa_rhs_adjust(self.h, self._ptr(self._field(rhs, "rhs"))))

    # -- explicit operators ------------------------------------------------------------
    def laplacian(self, x: Tensor, edge: bool, out: Tensor | None = None) -> Tensor:
        x = self._field(x, "laplacian")
        y = torch.empty_like(x) if out is None else self._field(out, "laplacian out")
        self._rc(self.lib.pa_laplacian(self.h, self._ptr(x), self._ptr(y), int(edge)))
        return y

    def grad(self, x: Tensor, edge: bool, out: Tensor | None = None) -> Tensor:
        x = self._field(x, "grad")
        y = out if out is not None else torch.empty((self.mesh.dim, *self.mesh.nx), dtype=self.dtype,
                                                    device=self.device)
        assert y.is_contiguous() and y.numel() == self.mesh.dim * self.mesh.N
        self._rc(self.lib.pa_grad(self.h, self._ptr(x), self._ptr(y), int(edge)))
        return y

    def div(self, kind: int, u: float | Tensor, x: Tensor, out: Tensor | None = None) -> Tensor:
        x = self._field(x, "div")
        y = torch.empty_like(x) if out is None else self._field(out, "div out")
        uf = None
        us = 0.0
        if isinstance(u, Tensor):
            uf = self._field(u if u.dim() == self.mesh.dim else u[0], "advection tensor")
        else:
            us = float(u)
        self._rc(self.lib.pa_div(self.h, kind, us, self._ptr(uf), self._ptr(x), self._ptr(y)))
        return y

    def div_edge(self, u: float | Tensor, x: Tensor, y: Tensor) -> None:
        uf, us = None, 0.0
        if isinstance(u, Tensor):
            uf = self._field(u if u.dim() == self.mesh.dim else u[0], "advection tensor")
        else:
            us = float(u)
        self._rc(self.lib.pa_div_edge(self.h, us, self._ptr(uf), self._ptr(self._field(x, "div edge")),
                                      self._ptr(self._field(y, "div edge out"))))

    def div_general(self, kind: int, edge: bool, x: Sequence[Tensor], u: float, u_int: Sequence[Tensor | None],
                    u_edge: Sequence[Tensor | None], out: Tensor) -> Tensor:
        """General Div (one entry per mesh axis in x / u_int / u_edge; None = the scalar u)."""
        sp = L.PaDivSpec()
        keep = []
        for a in range(self.mesh.dim):
            sp.x[a] = self._field(x[a], "div target").data_ptr()
            for name, lst in (("u_int", u_int), ("u_edge", u_edge)):
                t = lst[a]
                if t is not None:
                    t = self._field(t, "advection tensor")
                    keep.append(t)
                    getattr(sp, name)[a] = t.data_ptr()
        sp.u, sp.kind, sp.edge = float(u), int(kind), int(bool(edge))
        self._rc(self.lib.pa_div_general(self.h, C.byref(sp), self._ptr(self._field(out, "div out"))))
        return out

    def diff_flux(self, D: Sequence[Tensor], J: Sequence[Tensor], out: Tensor) -> Tensor:
        nd = self.mesh.dim
        Dp = (C.c_void_p * (nd * nd))(*[self._field(t, "diffusion tensor").data_ptr() for t in D])
        Jp = (C.c_void_p * nd)(*[self._field(t, "jacobian").data_ptr() for t in J])
        require_gpu(out, "DiffFlux out")
        assert out.is_contiguous() and out.numel() == nd * self.mesh.N and out.dtype == self.dtype
        self._rc(self.lib.pa_diff_flux(self.h, Dp, Jp, self._ptr(out)))
        return out

    def rfp_friction(self, Hr: Tensor, Hz: Tensor, pdf: Tensor) -> Tensor:
        out = torch.empty(tuple(self.mesh.nx), dtype=self.dtype, device=self.device)
        self._rc(self.lib.pa_rfp_friction(self.h, self._ptr(self._field(Hr, "H_r")), self._ptr(self._field(Hz, "H_z")),
                                          self._ptr(self._field(pdf, "pdf")), self._ptr(out)))
        return out

    def rfp_diffusion(self, Drr: Tensor, Drz: Tensor, Dzz: Tensor, pdf: Tensor) -> Tensor:
        out = torch.empty(tuple(self.mesh.nx), dtype=self.dtype, device=self.device)
        self._rc(self.lib.pa_rfp_diffusion(self.h, self._ptr(self._field(Drr, "D_rr")),
                                           self._ptr(self._field(Drz, "D_rz")), self._ptr(self._field(Dzz, "D_zz")),
                                           self._ptr(self._field(pdf, "pdf")), self._ptr(out)))
        return out

    def limiter(self, which: int, a: Tensor, b: Tensor) -> Tensor:
        require_gpu(a, "limiter")
        require_gpu(b, "limiter")
        assert a.shape == b.shape and a.dtype == b.dtype == self.dtype
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        self._rc(self.lib.pa_limiter(self.h, int(which), self._ptr(a), self._ptr(b), self._ptr(out), a.numel()))
        return out

    def euler_march(self, phi: Tensor, tmp: Tensor, kind: int, u: float | Tensor, nu: float, dt: float,
                    nsteps: int) -> Tensor:
        """``nsteps`` explicit Euler steps enqueued back to back, ping-ponging phi <-> tmp; returns the
        tensor that holds the final state."""
        phi = self._field(phi, "euler_march")
        tmp = self._field(tmp, "euler_march")
        uf, us = None, 0.0
        if isinstance(u, Tensor):
            uf = self._field(u if u.dim() == self.mesh.dim else u[0], "advection tensor")
        else:
            us = float(u)
        self._rc(self.lib.pa_euler_march(self.h, self._ptr(phi), self._ptr(tmp), kind, us, self._ptr(uf),
                                         float(nu), float(dt), int(nsteps)))
        return phi if nsteps % 2 == 0 else tmp

    def euler_step(self, phi: Tensor, out: Tensor, kind: int, u: float | Tensor, nu: float, dt: float) -> None:
        phi = self._field(phi, "euler_step")
        out = self._field(out, "euler_step")
        uf, us = None, 0.0
        if isinstance(u, Tensor):
            uf = self._field(u if u.dim() == self.mesh.dim else u[0], "advection tensor")
        else:
            us = float(u)
        self._rc(self.lib.pa_euler_step(self.h, self._ptr(phi), self._ptr(out), kind, us, self._ptr(uf),
                                        float(nu), float(dt)))

    # -- solvers --------------------------------------------------------------------------
    def keep_old(self, x_old: Tensor | None) -> None:
        """Buffer that receives the iterate before the last executed solver iteration (Field.VARo), or None."""
        self._keep["x_old"] = x_old
        self._rc(self.lib.pa_solver_keep_old(self.h, self._ptr(None if x_old is None else self._field(x_old, "x_old"))))

    def solve(self, method: str, x: Tensor, rhs: Tensor, tol: float, max_it: int,
              omega: float = 1.0) -> L.PaReport:
        x = self._field(x, "solve")
        rhs = self._field(rhs, "solve rhs")
        rep = L.PaReport()
        if method == "cg":
            rc = self.lib.pa_cg(self.h, self._ptr(x), self._ptr(rhs), float(tol), int(max_it), C.byref(rep))
        elif method == "bicgstab":
            rc = self.lib.pa_bicgstab(self.h, self._ptr(x), self._ptr(rhs), float(tol), int(max_it), C.byref(rep))
        elif method == "jacobi":
            rc = self.lib.pa_jacobi(self.h, self._ptr(x), self._ptr(rhs), float(tol), int(max_it), float(omega),
                                    C.byref(rep))
        else:
            raise RuntimeError(f"unknown method {method}")
        if rc == L.PA_E_NONFINITE:
            # linalg.py:334-336
            raise RuntimeError(f"Invalid tolerance detected! tol: {rep.tol}")
        self._rc(rc)
        return rep

    # stepwise CG (bench.py and the slab driver)
    def cg_begin(self, x: Tensor, rhs: Tensor, tol: float, max_it: int) -> None:
        self._keep["cg"] = (x, rhs)
        self._rc(self.lib.pa_cg_begin(self.h, self._ptr(self._field(x, "cg")), self._ptr(self._field(rhs, "cg rhs")),
                                      float(tol), int(max_it)))

    def cg_iterate(self, n: int) -> None:
        self._rc(self.lib.pa_cg_iterate(self.h, int(n)))

    def cg_phase_a(self) -> None:
        self._rc(self.lib.pa_cg_phase_a(self.h))

    def cg_phase_b(self) -> None:
        self._rc(self.lib.pa_cg_phase_b(self.h))

    def cg_bc(self) -> None:
        self._rc(self.lib.pa_cg_bc(self.h))

    def cg_finish_iter(self) -> None:
        self._rc(self.lib.pa_cg_finish_iter(self.h))

    # stepwise BiCGSTAB on a slab (pyapes_amd/slab.py SlabBiCGSTAB)
    def slab_set_v(self, v_send_lo: Tensor | None, v_send_hi: Tensor | None, v_recv_lo: Tensor | None,
                   v_recv_hi: Tensor | None) -> None:
        self._keep["slab_v"] = (v_send_lo, v_send_hi, v_recv_lo, v_recv_hi)
        self._rc(self.lib.pa_slab_set_v(self.h, self._ptr(v_send_lo), self._ptr(v_send_hi), self._ptr(v_recv_lo),
                                        self._ptr(v_recv_hi)))

    def bicg_begin(self, x: Tensor, rhs: Tensor, tol: float, max_it: int) -> None:
        self._keep["cg"] = (x, rhs)
        self._rc(self.lib.pa_bicg_begin(self.h, self._ptr(self._field(x, "bicgstab")),
                                        self._ptr(self._field(rhs, "bicgstab rhs")), float(tol), int(max_it)))

    def bicg_start(self) -> None:
        self._rc(self.lib.pa_bicg_start(self.h))

    def bicg_pv(self) -> None:
        self._rc(self.lib.pa_bicg_pv(self.h))

    def bicg_st(self) -> None:
        self._rc(self.lib.pa_bicg_st(self.h))

    def bicg_x(self) -> None:
        self._rc(self.lib.pa_bicg_x(self.h))

    def bicg_bc(self) -> None:
        self._rc(self.lib.pa_bicg_bc(self.h))

    def bicg_finish(self) -> None:
        self._rc(self.lib.pa_bicg_finish(self.h))

    def bicg_end(self) -> L.PaReport:
        rep = L.PaReport()
        rc = self.lib.pa_bicg_end(self.h, C.byref(rep))
        self._keep.pop("cg", None)
        if rc == L.PA_E_NONFINITE:
            raise RuntimeError(f"Invalid tolerance detected! tol: {rep.tol}")
        self._rc(rc)
        return rep

    def resident_plan(self, method: str = "cg") -> tuple[int, tuple[int, int, int]]:
        """(workgroups, boxes per internal axis) a solve with `method` on the bound mesh, BCs and equation would
        run resident with; (0, (0, 0, 0)) when the launch-per-phase loops would run."""
        boxes = (C.c_int * 3)()
        g = self.lib.pa_resident_plan(self.h, {"cg": 0, "jacobi": 1, "bicgstab": 2}[method.lower()], boxes)
        if g < 0:
            self._rc(g)
        return int(g), (int(boxes[0]), int(boxes[1]), int(boxes[2]))

    def resident_used(self) -> int:
        """Workgroups of the last solve's resident launch (0: the launch-per-phase loops ran)."""
        return int(self.lib.pa_resident_used(self.h))

    def cg_abort(self) -> None:
        self._rc(self.lib.pa_cg_abort(self.h))
        self._keep.pop("cg", None)

    def cg_end(self) -> L.PaReport:
        rep = L.PaReport()
        rc = self.lib.pa_cg_end(self.h, C.byref(rep))
        self._keep.pop("cg", None)
        if rc == L.PA_E_NONFINITE:
            raise RuntimeError(f"Invalid tolerance detected! tol: {rep.tol}")
        self._rc(rc)
        return rep

    def report(self) -> L.PaReport:
        rep = L.PaReport()
        self._rc(self.lib.pa_report_read(self.h, C.byref(rep)))
        return rep

    def scalars(self) -> dict[str, float]:
        """alpha, beta, rho, omega ... of the last executed solver iteration (pa_scalars_read)"""
        v = (C.c_double * 16)()
        self._rc(self.lib.pa_scalars_read(self.h, v))
        names = ("alpha", "beta", "rr", "rr_old", "dAd", "tol", "rho", "omega", "rho_next", "r0v", "ts", "tt", "r0t", "itr")
        return {n: float(v[i]) for i, n in enumerate(names)}

    def place_stats(self) -> dict[str, float]:
        """Accounts of the online placement search of large CG solves (pa_place_stats)."""
        v = (C.c_double * 10)()
        self._rc(self.lib.pa_place_stats(self.h, v))
        names = ("state", "trials", "kept", "allocations", "spent_us", "timed_us", "best_pair_us", "blocks_held",
                 "first_pair_us", "passes_done")
        out = {n: float(v[i]) for i, n in enumerate(names)}
        out["state"] = {-1: "off", 0: "idle", 1: "searching", 2: "done"}[int(v[0])]
        return out

    def profile(self, on: bool) -> None:
        self._rc(self.lib.pa_profile_set(self.h, int(on)))

    def profile_read(self) -> dict[str, float]:
        ma, mb = C.c_double(), C.c_double()
        na, nb = C.c_int64(), C.c_int64()
        self._rc(self.lib.pa_profile_read(self.h, C.byref(ma), C.byref(na), C.byref(mb), C.byref(nb)))
        return {"phase_a_ms": ma.value / max(na.value, 1), "phase_a_n": na.value,
                "phase_b_ms": mb.value / max(nb.value, 1), "phase_b_n": nb.value}

    # -- RCCL inside the library (slab iterations without host work between the phases) -------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        rc = self.lib.pa_comm_unique_id(C.cast(buf, C.c_void_p))
        if rc != L.PA_OK:
            raise L.PaError(rc, "pa_comm_unique_id: librccl not available")
        return buf.raw

    def comm_available(self) -> bool:
        return bool(self.lib.pa_comm_available())

    def comm_size(self) -> int:
        n = C.c_int(0)
        self._rc(self.lib.pa_comm_count(self.h, C.byref(n)))
        return int(n.value)

    def comm_init(self, rank: int, world: int, uid: bytes) -> None:
        buf = C.create_string_buffer(uid, 128)
        self._rc(self.lib.pa_comm_init(self.h, int(rank), int(world), C.cast(buf, C.c_void_p)))

    def comm_selftest(self, timeout_s: float = 20.0) -> None:
        self._rc(self.lib.pa_comm_selftest(self.h, float(timeout_s)))

    def comm_plan(self, nb_lo: int | None, nb_hi: int | None, send_lo: Tensor | None, send_hi: Tensor | None,
                  recv_lo: Tensor | None, recv_hi: Tensor | None) -> None:
        p = L.PaExchange()
        p.nb_lo = -1 if nb_lo is None else int(nb_lo)
        p.nb_hi = -1 if nb_hi is None else int(nb_hi)
        for name, t in (("send_lo", send_lo), ("send_hi", send_hi), ("recv_lo", recv_lo), ("recv_hi", recv_hi)):
            if t is not None:
                require_gpu(t, "exchange buffer")
                assert t.is_contiguous() and t.dtype == self.dtype
            setattr(p, name, None if t is None else t.data_ptr())
            setattr(p, "n_" + name, 0 if t is None else t.numel())
        self._keep["plan"] = (p, send_lo, send_hi, recv_lo, recv_hi)
        self._rc(self.lib.pa_comm_plan(self.h, C.byref(p)))

    def cg_fold_plan(self) -> list[int]:
        rows = (C.c_int64 * 3)()
        self._rc(self.lib.pa_cg_fold_plan(self.h, rows))
        return [int(v) for v in rows]

    # vector steps of the host-stepped loops (solver/host_stepped.py)
    def vec_axpy(self, out: Tensor, y: Tensor, a: float, x: Tensor) -> Tensor:
        """out = y + a * x (the product rounded first); out may alias y or x"""
        self._rc(self.lib.pa_vec_axpy(self.h, self._ptr(self._field(out, "vec_axpy")), self._ptr(self._field(y, "vec_axpy")),
                                      float(a), self._ptr(self._field(x, "vec_axpy"))))
        return out

    def vec_dot(self, a: Tensor, b: Tensor, diff: bool = False) -> float:
        """sum a.b, or sum (a - b)^2 with ``diff`` (synchronises)"""
        res = C.c_double(0.0)
        self._rc(self.lib.pa_vec_dot(self.h, self._ptr(self._field(a, "vec_dot")), self._ptr(self._field(b, "vec_dot")),
                                     1 if diff else 0, C.byref(res)))
        return float(res.value)

    def vec_mask_interior(self, x: Tensor) -> Tensor:
        """x <- 0 off the interior set of the BC list bound last"""
        self._rc(self.lib.pa_vec_mask_interior(self.h, self._ptr(self._field(x, "vec_mask_interior"))))
        return x

    # stepwise Jacobi on a slab (pyapes_amd/slab.py SlabJacobi)
    def jacobi_begin(self, x: Tensor, rhs: Tensor, tol: float, max_it: int, omega: float = 1.0) -> None:
        self._keep["cg"] = (x, rhs)
        self._rc(self.lib.pa_jacobi_begin(self.h, self._ptr(self._field(x, "jacobi")),
                                          self._ptr(self._field(rhs, "jacobi rhs")), float(tol), int(max_it), float(omega)))

    def jacobi_sweep(self) -> None:
        self._rc(self.lib.pa_jacobi_sweep(self.h))

    def jacobi_bc(self) -> None:
        self._rc(self.lib.pa_jacobi_bc(self.h))

    def jacobi_finish(self) -> None:
        self._rc(self.lib.pa_jacobi_finish(self.h))

    def jacobi_end(self) -> L.PaReport:
        rep = L.PaReport()
        rc = self.lib.pa_jacobi_end(self.h, C.byref(rep))
        self._keep.pop("cg", None)
        if rc == L.PA_E_NONFINITE:
            raise RuntimeError(f"Invalid tolerance detected! tol: {rep.tol}")
        self._rc(rc)
        return rep

    def jacobi_iterate_comm(self, n: int) -> None:
        self._rc(self.lib.pa_jacobi_iterate_comm(self.h, int(n)))

    def cg_fold_set(self, rows: Sequence[int]) -> None:
        arr = (C.c_int64 * 3)(*[int(v) for v in rows])
        self._rc(self.lib.pa_cg_fold_set(self.h, arr))

    def cg_iterate_comm(self, n: int) -> None:
        self._rc(self.lib.pa_cg_iterate_comm(self.h, int(n)))

    def bicg_iterate_comm(self, n: int) -> None:
        self._rc(self.lib.pa_bicg_iterate_comm(self.h, int(n)))

    def comm_destroy(self) -> None:
        self._rc(self.lib.pa_comm_destroy(self.h))
        self._keep.pop("plan", None)

    def comm_abort(self) -> None:
        """Abort the library's communicators (a collective that some rank never joined) and drain the streams."""
        self._rc(self.lib.pa_comm_abort(self.h))
        self._keep.pop("plan", None)
        self.comm_ready = None

    def comm_overlap(self) -> bool:
        """True when the plane exchange runs on the library's second communicator + stream."""
        return bool(self.lib.pa_comm_overlap(self.h))

    def comm_impl(self) -> str:
        return self.lib.pa_comm_impl().decode()

    def stream_wait(self, timeout_s: float) -> bool:
        """True once everything enqueued on the ctx stream(s) has run; False if work is still queued after
        ``timeout_s`` seconds."""
        return self.lib.pa_stream_wait(self.h, float(timeout_s)) == L.PA_OK

    def slab_set(self, bufs: dict[str, Tensor | None] | None) -> None:
        if bufs is None:
            self._rc(self.lib.pa_slab_set(self.h, None))
            self._keep.pop("slab", None)
            return
        s = L.PaSlab()
        for name, _ in L.PaSlab._fields_:
            t = bufs.get(name)
            setattr(s, name, None if t is None else t.data_ptr())
        self._keep["slab"] = (s, bufs)
        self._rc(self.lib.pa_slab_set(self.h, C.byref(s)))


def context_for(mesh: Any) -> HipContext:
    """The (lazily created) context of a mesh."""
    if getattr(mesh, "_hip", None) is None:
        mesh._hip = HipContext(mesh)
    return mesh._hip
