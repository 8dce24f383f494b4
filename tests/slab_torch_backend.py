"""TEST INFRASTRUCTURE: a torch-CPU stand-in for the stepwise C-ABI calls that
``pyapes_amd.slab.SlabCG`` / ``SlabBiCGSTAB`` / ``SlabJacobi`` / ``SlabEuler`` drive, so that the driver's communication pattern (ghost planes,
ring wrap, periodic far planes, all-reduce slices, call order) can be exercised with gloo and
world_size 2 on a machine without a GPU.  It follows SURVEY Appendix A on the LOCAL slab with
ghost planes; results are checked against the single-domain oracle.  Never imported by the
product."""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch


class TorchSlabBackend:
    is_standin = True      # (pyapes_amd skips its "needs the GPU" guard for a backend object that says so)

    def __init__(self, mesh):
        self.mesh = mesh
        self.off = mesh.i_off
        self.gn = list(mesh.global_nx)
        self.n = list(mesh.nx)
        self.f = mesh.dtype.float
        self.h = [torch.tensor(v, dtype=self.f) for v in mesh.dx_list]
        self.bufs = None
        self.itr = 0

    # -- configuration ----------------------------------------------------------------
    def slab_set(self, bufs):
        self.bufs = bufs

    def bind_bcs(self, var, bcs, comp, for_rhs=False):
        self.bcs = [(bc.bc_face, bc.bc_type, bc.bc_val) for bc in bcs]
        self.types = {f: t for f, t, _ in self.bcs}

    def set_terms(self, terms):
        assert len(terms) == 1 and terms[0]["kind"] == 0
        self.sign = torch.tensor(float(terms[0].get("sign", 1.0)), dtype=self.f)
        c = terms[0].get("coeff")
        self.coeff = None if c is None else torch.tensor(float(c), dtype=self.f)

    def _treat(self, face):
        return self.types.get(face) in ("neumann", "symmetry")

    def _coef_vec(self, axis, idx_global, N, lo_face, hi_face):
        """(cP, cC, cM) vectors over the given global indices (fdc.py:376-423)."""
        h2 = self.h[axis] * self.h[axis]
        inv = torch.tensor(1.0, dtype=self.f) / h2
        m2 = torch.tensor(-2.0, dtype=self.f) / h2
        c23 = torch.tensor(2.0 / 3.0, dtype=self.f) / h2
        cP = inv.repeat(len(idx_global)); cC = m2.repeat(len(idx_global)); cM = inv.repeat(len(idx_global))
        for q, g in enumerate(idx_global):
            if self._treat(lo_face) and g == 1:
                cP[q], cC[q], cM[q] = c23, -c23, 0.0
            if self._treat(hi_face) and g == N - 2:
                cP[q], cC[q], cM[q] = 0.0, -c23, c23
        return cP, cC, cM

    def _S(self):
        """boolean mask of the interior set on the local slab (mesh/tools.py:7-20)."""
        lim = []
        for a, (lo, hi) in enumerate((("xl", "xu"), ("yl", "yu"), ("zl", "zu"))):
            l = 0 if self.types.get(lo) == "periodic" else 1
            u = self.gn[a] - 1 if self.types.get(hi) == "periodic" else self.gn[a] - 2
            lim.append((l, u))
        gi = torch.arange(self.off, self.off + self.n[0])
        m0 = (gi >= lim[0][0]) & (gi <= lim[0][1])
        j = torch.arange(self.n[1]); m1 = (j >= lim[1][0]) & (j <= lim[1][1])
        k = torch.arange(self.n[2]); m2 = (k >= lim[2][0]) & (k <= lim[2][1])
        return m0[:, None, None] & m1[None, :, None] & m2[None, None, :]

    def _A(self, v, glo, ghi):
        """sign*coeff*laplacian on the local planes; ghost planes complete axis 0."""
        ax = self._lap(v, glo, ghi)
        if self.coeff is not None:
            ax = ax * self.coeff
        return ax * self.sign

    def _lap(self, v, glo, ghi):
        z = torch.zeros_like(v[0])
        pad = torch.cat([(glo if glo is not None else z)[None], v, (ghi if ghi is not None else z)[None]], 0)
        gi = list(range(self.off, self.off + self.n[0]))
        c0 = self._coef_vec(0, gi, self.gn[0], "xl", "xu")
        c1 = self._coef_vec(1, list(range(self.n[1])), self.gn[1], "yl", "yu")
        c2 = self._coef_vec(2, list(range(self.n[2])), self.gn[2], "zl", "zu")
        vp, vm = pad[2:], pad[:-2]
        t = c0[0][:, None, None] * vp
        t = t + c0[1][:, None, None] * v
        t = t + c0[2][:, None, None] * vm
        ax = t
        t = c1[0][None, :, None] * torch.roll(v, -1, 1)
        t = t + c1[1][None, :, None] * v
        t = t + c1[2][None, :, None] * torch.roll(v, 1, 1)
        ax = ax + t
        t = c2[0][None, None, :] * torch.roll(v, -1, 2)
        t = t + c2[1][None, None, :] * v
        t = t + c2[2][None, None, :] * torch.roll(v, 1, 2)
        ax = ax + t
        return ax

    def rhs_adjust(self, rhs):
        for face, typ, val in self.bcs:
            if typ != "neumann":
                continue
            a = "xyz".index(face[0]); side = face[1]
            N = self.gn[a]
            prev = 1 if side == "l" else N - 2
            nv = -1.0 if side == "l" else 1.0
            add = torch.tensor(2.0 / 3.0, dtype=self.f) * torch.tensor(float(val) * nv, dtype=self.f) / self.h[a]
            if a == 0:
                li = prev - self.off
                if 0 <= li < self.n[0]:
                    rhs[li] += add
            elif a == 1:
                rhs[:, prev, :] += add
            else:
                rhs[:, :, prev] += add

    # -- BC fill (bcs.py:200-280) on the local slab ----------------------------------------
    def apply_bc_bound(self, x):
        n0 = self.n[0]
        last = self.off + n0 == self.gn[0]
        first = self.off == 0
        c43 = torch.tensor(4 / 3, dtype=self.f); c13 = torch.tensor(1 / 3, dtype=self.f)
        for face, typ, val in self.bcs:
            a = "xyz".index(face[0]); lower = face[1] == "l"
            if a == 0 and ((lower and not first) or (not lower and not last)):
                continue
            sl = [slice(None)] * 3
            def at(i):
                s = list(sl); s[a] = i; return tuple(s)
            f_, p1, p2 = (0, 1, 2) if lower else (-1, -2, -3)
            if typ == "dirichlet":
                x[at(f_)] = float(val)
            elif typ == "neumann":
                g = self.mesh._gx_host[a]
                dxf = (g[0] - g[1]) if lower else (g[-1] - g[-2])
                ct = torch.tensor((2.0 / 3.0) * float(val), dtype=self.f) * dxf * (-1.0 if lower else 1.0)
                x[at(f_)] = c43 * x[at(p1)] - c13 * x[at(p2)] + ct
            elif typ == "symmetry":
                x[at(f_)] = x[at(p1)]
            elif typ == "periodic":
                if a == 0 and self.n[0] != self.gn[0]:
                    if lower:
                        x[0] = x[1] - self.bufs["bc_far_lo0"] + self.bufs["bc_far_lo1"]
                    else:
                        x[-1] = self.bufs["bc_far_hi0"] - x[-1] + x[-2]
                elif lower:
                    x[at(0)] = x[at(1)] - x[at(-1)] + x[at(-2)]
                else:
                    x[at(-1)] = x[at(0)]

    # -- stepwise CG (linalg.py:74-159 split at its two reductions) ---------------------------
    def cg_begin(self, x, rhs, tol, max_it):
        b = self.bufs
        self.x, self.tolerance, self.max_it = x, tol, max_it
        self.S = self._S()
        ax = self._A(x, b["x_ghost_lo"], b["x_ghost_hi"])
        self.r = torch.where(self.S, rhs - ax, torch.zeros_like(x))
        self.d = self.r.clone()
        self.dg = [torch.zeros_like(x[0]), torch.zeros_like(x[0])]
        self.beta = torch.tensor(0.0, dtype=self.f)
        self.itr, self.done, self.tol = 0, not (1.0 > tol), 1.0
        self._send_r()
        b["sums"][1] = float(torch.sum(self.r * self.r))
        self.first = True

    def _send_r(self):
        b = self.bufs
        if b["r_send_lo"] is not None:
            b["r_send_lo"].copy_(self.r[0])
        if b["r_send_hi"] is not None:
            b["r_send_hi"].copy_(self.r[-1])

    def cg_phase_a(self):
        if self.done:
            return
        b = self.bufs
        if self.first:
            self.rr = torch.tensor(float(b["sums"][1]), dtype=self.f)
            self.first = False
        rl, rh = b["r_recv_lo"], b["r_recv_hi"]
        glo = None if rl is None else rl + self.beta * self.dg[0]
        ghi = None if rh is None else rh + self.beta * self.dg[1]
        self.d = torch.where(self.S, self.r + self.beta * self.d, torch.zeros_like(self.d))
        if glo is not None:
            self.dg[0] = glo
        if ghi is not None:
            self.dg[1] = ghi
        self.Ad = torch.where(self.S, self._A(self.d, glo, ghi), torch.zeros_like(self.d))
        b["sums"][0] = float(torch.sum(self.d * self.Ad))

    def cg_phase_b(self):
        if self.done:
            return
        b = self.bufs
        dAd = torch.tensor(float(b["sums"][0]), dtype=self.f)
        a = self.rr / dAd
        self.alpha = torch.nan_to_num(a, nan=0.0, posinf=0.0, neginf=0.0)
        self.x_old = self.x.clone()
        self.x.copy_(torch.where(self.S, self.x + self.alpha * self.d, self.x))
        self.r = torch.where(self.S, self.r - self.alpha * self.Ad, torch.zeros_like(self.r))
        self._send_r()
        for key, plane in (("x_pack_lo1", 1), ("x_pack_hi0", -1), ("x_pack_hi1", -2)):
            if b.get(key) is not None:
                b[key].copy_(self.x[plane])

    def cg_bc(self):
        if self.done:
            return
        b = self.bufs
        self.apply_bc_bound(self.x)
        b["sums"][1] = float(torch.sum(self.r * self.r))
        df = self.x - self.x_old
        b["sums"][2] = float(torch.sum(df * df))

    def cg_finish_iter(self):
        if self.done:
            return
        b = self.bufs
        rr_new = torch.tensor(float(b["sums"][1]), dtype=self.f)
        self.tol = float(torch.sqrt(torch.tensor(float(b["sums"][2]), dtype=self.f)))
        if math.isnan(self.tol) or math.isinf(self.tol):
            raise RuntimeError("Invalid tolerance detected!")
        self.beta = rr_new / self.rr
        self.rr = rr_new
        self.itr += 1
        if self.itr > self.max_it or not (self.tol > self.tolerance):
            self.done = True

    def report(self):
        return SimpleNamespace(itr=self.itr, tol=self.tol, converge=self.itr < self.max_it, status=0)

    def cg_end(self):
        return self.report()

    # -- stepwise BiCGSTAB (linalg.py:162-279 split at its reductions and exchanges; include/pyapes_hip.h) -------
    def slab_set_v(self, v_send_lo, v_send_hi, v_recv_lo, v_recv_hi):
        self.vb = {"send_lo": v_send_lo, "send_hi": v_send_hi, "recv_lo": v_recv_lo, "recv_hi": v_recv_hi}

    def _t(self, v):
        return torch.tensor(float(v), dtype=self.f)

    def bicg_begin(self, x, rhs, tol, max_it):
        b = self.bufs
        self.x, self.tolerance, self.max_it = x, tol, max_it
        self.S = self._S()
        ax = self._A(x, b["x_ghost_lo"], b["x_ghost_hi"])
        z = torch.zeros_like(x)
        self.r0 = torch.where(self.S, rhs - ax, z)
        self.r = self.r0.clone()
        self.p, self.v, self.s, self.t = z.clone(), z.clone(), z.clone(), z.clone()
        self.pg = [torch.zeros_like(x[0]), torch.zeros_like(x[0])]
        self.itr, self.done, self.fe, self.tol = 0, False, False, 1.0
        self._send_r()
        b["sums"][1] = float(torch.sum(self.r0 * self.r0))

    def bicg_start(self):
        one = self._t(1.0)
        self.rho_next = self._t(self.bufs["sums"][1])
        self.tol = float(torch.sqrt(self.rho_next))
        self.rho, self.alpha, self.omega = one, one, one
        self.beta = self.rho_next / self.rho * self.alpha / self.omega
        self.rho = self.rho_next

    def bicg_pv(self):
        if self.done:
            return
        b, vb = self.bufs, self.vb
        rl, rh = b["r_recv_lo"], b["r_recv_hi"]
        glo = None if rl is None else rl + self.beta * (self.pg[0] - self.omega * vb["recv_lo"])
        ghi = None if rh is None else rh + self.beta * (self.pg[1] - self.omega * vb["recv_hi"])
        self.p = self.r + self.beta * (self.p - self.omega * self.v)
        if glo is not None:
            self.pg[0] = glo
        if ghi is not None:
            self.pg[1] = ghi
        self.v = torch.where(self.S, self._A(self.p, glo, ghi), torch.zeros_like(self.p))
        b["sums"][0] = float(torch.sum(self.r0 * self.v))
        if vb["send_lo"] is not None:
            vb["send_lo"].copy_(self.v[0])
        if vb["send_hi"] is not None:
            vb["send_hi"].copy_(self.v[-1])

    def bicg_st(self):
        if self.done:
            return
        b, vb = self.bufs, self.vb
        self.itr += 1
        self.alpha = torch.nan_to_num(self.rho / self._t(b["sums"][0]), nan=0.0, posinf=0.0, neginf=0.0)
        self.s = self.r - self.alpha * self.v
        rl, rh = b["r_recv_lo"], b["r_recv_hi"]
        sglo = None if rl is None else rl - self.alpha * vb["recv_lo"]
        sghi = None if rh is None else rh - self.alpha * vb["recv_hi"]
        b["sums"][1] = float(torch.sum(self.s * self.s))
        self.t = torch.where(self.S, self._A(self.s, sglo, sghi), torch.zeros_like(self.s))
        b["sums"][2] = float(torch.sum(self.t * self.s))
        b["sums"][3] = float(torch.sum(self.t * self.t))
        b["sums"][4] = float(torch.sum(self.r0 * self.t))

    def bicg_x(self):
        if self.done:
            return
        b = self.bufs
        self.tol = float(torch.sqrt(self._t(b["sums"][1])))
        if math.isnan(self.tol) or math.isinf(self.tol):
            raise RuntimeError("Invalid tolerance detected!")
        self.fe = self.tol <= self.tolerance
        if self.fe:
            self.x.copy_(self.x + self.alpha * self.p)
        else:
            self.omega = torch.nan_to_num(self._t(b["sums"][2]) / self._t(b["sums"][3]), nan=0.0, posinf=0.0, neginf=0.0)
            self.rho_next = -self.omega * self._t(b["sums"][4])
            self.x.copy_(self.x + self.alpha * self.p + self.s * self.omega)
            self.r = self.s - self.omega * self.t
        self._send_r()
        for key, plane in (("x_pack_lo1", 1), ("x_pack_hi0", -1), ("x_pack_hi1", -2)):
            if b.get(key) is not None:
                b[key].copy_(self.x[plane])

    def bicg_bc(self):
        if self.done:
            return
        self.apply_bc_bound(self.x)
        self.bufs["sums"][5] = float(torch.sum(self.r * self.r))

    def bicg_finish(self):
        if self.done:
            return
        if self.fe:
            self.done = True
            return
        self.tol = float(torch.sqrt(self._t(self.bufs["sums"][5])))
        if math.isnan(self.tol) or math.isinf(self.tol):
            raise RuntimeError("Invalid tolerance detected!")
        if self.tol <= self.tolerance or self.itr >= self.max_it:
            self.done = True
        self.beta = self.rho_next / self.rho * self.alpha / self.omega
        self.rho = self.rho_next

    def bicg_end(self):
        return self.report()

    # -- stepwise Jacobi (the sweep of pa_jacobi [new, SURVEY a15] split at its exchanges; include/pyapes_hip.h) -------
    def _diag(self):
        gi = list(range(self.off, self.off + self.n[0]))
        c0 = self._coef_vec(0, gi, self.gn[0], "xl", "xu")[1]
        c1 = self._coef_vec(1, list(range(self.n[1])), self.gn[1], "yl", "yu")[1]
        c2 = self._coef_vec(2, list(range(self.n[2])), self.gn[2], "zl", "zu")[1]
        dg = c0[:, None, None] + c1[None, :, None]
        dg = dg + c2[None, None, :]
        if self.coeff is not None:
            dg = dg * self.coeff
        return dg * self.sign

    def jacobi_begin(self, x, rhs, tol, max_it, omega=1.0):
        self.x, self.rhs_j, self.tolerance, self.max_it = x, rhs, tol, max_it
        self.om = self._t(omega)
        self.S = self._S()
        self.dg = self._diag()
        self.itr, self.done, self.tol = 0, False, 1.0

    def jacobi_sweep(self):
        if self.done:
            return
        b = self.bufs
        ax = self._A(self.x, b["x_ghost_lo"], b["x_ghost_hi"])
        self.x_old = self.x.clone()
        w = self.om * ((self.rhs_j - ax) / self.dg)
        self.x.copy_(torch.where(self.S, self.x + w, self.x))
        for key, plane in (("x_pack_lo1", 1), ("x_pack_hi0", -1), ("x_pack_hi1", -2)):
            if b.get(key) is not None:
                b[key].copy_(self.x[plane])

    def jacobi_bc(self):
        if self.done:
            return
        b = self.bufs
        self.apply_bc_bound(self.x)
        df = self.x - self.x_old
        b["sums"][2] = float(torch.sum(df * df))
        if b["r_send_lo"] is not None:
            b["r_send_lo"].copy_(self.x[0])
        if b["r_send_hi"] is not None:
            b["r_send_hi"].copy_(self.x[-1])

    def jacobi_finish(self):
        if self.done:
            return
        self.tol = float(torch.sqrt(self._t(self.bufs["sums"][2])))
        if math.isnan(self.tol) or math.isinf(self.tol):
            raise RuntimeError("Invalid tolerance detected!")
        self.itr += 1
        if self.itr > self.max_it or not (self.tol > self.tolerance):
            self.done = True

    def jacobi_end(self):
        return self.report()

    # -- explicit Euler step on a slab (solver/march.py; oracle euler_step / div_upwind_intended): kernel only, no fill ----
    def euler_step(self, phi, out, kind, u, nu, dt):
        assert not isinstance(u, torch.Tensor), "stand-in: scalar advection speed"
        b = self.bufs
        glo, ghi = b["x_ghost_lo"], b["x_ghost_hi"]
        lap = self._lap(phi, glo, ghi)
        z = torch.zeros_like(phi[0])
        pad = torch.cat([(glo if glo is not None else z)[None], phi, (ghi if ghi is not None else z)[None]], 0)
        uu = self._t(u)
        up, um = torch.clamp(uu, min=0.0), torch.clamp(uu, max=0.0)
        adv = torch.zeros_like(phi)
        for a in range(3):
            inv = torch.ones((), dtype=self.f) / self.h[a]
            if a == 0:
                bwd, fwd = phi - pad[:-2], pad[2:] - phi
            else:
                bwd, fwd = phi - torch.roll(phi, 1, a), torch.roll(phi, -1, a) - phi
            adv = adv + (up * bwd + um * fwd) * inv
        new = phi + self._t(dt) * (self._t(nu) * lap - adv)
        out.copy_(torch.where(self._S(), new, phi))
