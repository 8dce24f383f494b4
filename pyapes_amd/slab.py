"""Slab decomposition along axis 0 and the multi-GPU CG / BiCGSTAB / Jacobi drivers and the explicit Euler march (new).

The reference is single-device.  Here the 3-D grid is cut into P slabs of whole
(n1 x n2) planes, one per rank / GPU.  Per CG iteration the ranks exchange

  * one boundary plane of the residual r with each axis-0 neighbour (ring wrap when
    axis 0 is periodic) -- the ghost planes of the search direction are NOT sent: each
    rank advances them with the same recurrence d' = r + beta d the owner uses, bit for bit;
  * when axis 0 is periodic, the three planes of x its BC fill reads across the ring
    (x[N-1], x[N-2] to the lower end rank, x[1] to the upper end rank);
  * two all-reduces of scalars (sum d.Ad, then sum r.r + the stop-test sum).

Everything else is rank-local.  Communication goes through ``torch.distributed``
(backend "nccl" = RCCL over xGMI on the GPUs; "gloo" in the CPU tests), compute through
a *backend* object with the stepwise C-ABI calls (``HipContext`` in the product; the tests
substitute an oracle-backed stand-in to exercise this driver on CPU with world_size 2).
"""
from __future__ import annotations

from typing import Any, Sequence

import torch
from torch import Tensor


def slab_extent(n0: int, rank: int, world: int) -> tuple[int, int]:
    """-> (first global plane, number of planes) of ``rank``; remainders go to the low ranks."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"slab_extent: bad rank/world {rank}/{world}")
    if n0 < 3 * world:
        raise ValueError(f"slab_extent: {n0} planes are too few for {world} slabs (>= 3 planes each)")
    base, rem = divmod(n0, world)
    off = rank * base + min(rank, rem)
    return off, base + (1 if rank < rem else 0)


class _SlabDriver:
    """What the slab solvers share: the neighbour relation, the packed exchange buffers and the plane exchange /
    all-reduce primitives over ``torch.distributed``."""

    uses_lib_comm = False     # (SlabCG: RCCL inside the library where the process group allows it)

    def __init__(self, mesh: Any, var: Any, rhs: Tensor, terms: Sequence[dict], dist: Any,
                 backend: Any = None, group: Any = None):
        assert mesh.slab is not None, f"{type(self).__name__} needs a Mesh(..., slab=(rank, world))"
        self.mesh, self.var, self.dist, self.group = mesh, var, dist, group
        self.rank, self.world = mesh.slab
        if backend is None:
            from .hip.context import context_for
            backend = context_for(mesh)
        self.be = backend
        types = {bc.bc_face: bc.bc_type for bc in var.bcs}
        self.periodic0 = types.get("xl") == "periodic" or types.get("xu") == "periodic"
        if self.periodic0 and not (types.get("xl") == "periodic" and types.get("xu") == "periodic"):
            raise ValueError("slab solvers: axis 0 must be periodic on both faces or on none")
        order = [bc.bc_face for bc in var.bcs]
        if order[:2] != ["xl", "xu"]:
            raise ValueError("slab solvers: the BC list must start with xl, xu (factory order)")
        P, r = self.world, self.rank
        self.nb_lo = r - 1 if r > 0 else (P - 1 if self.periodic0 else None)
        self.nb_hi = r + 1 if r < P - 1 else (0 if self.periodic0 else None)
        self.x = var()[0]
        self.rhs = rhs[0] if rhs.dim() == 4 else rhs
        dev, f = self.x.device, self.x.dtype
        plane = tuple(self.x.shape[1:])

        def buf(cond=True):
            return torch.zeros(plane, dtype=f, device=dev) if cond else None

        self.sums = torch.zeros(8, dtype=torch.float64, device=dev)
        # Packed exchange buffers, one message per neighbour and iteration:
        #   down (to nb_lo):  [ r first plane | x[1] if I am the lower end of a periodic ring ]
        #   up   (to nb_hi):  [ r last plane  | x[n0-1], x[n0-2] if I am the upper end of the ring ]
        # and the mirror-image receive buffers; the C side gets plane views of them.
        first, last = self.periodic0 and r == 0, self.periodic0 and r == P - 1
        self.k_lo, self.k_hi = 1 + (1 if first else 0), 1 + (2 if last else 0)   # planes I send down / up
        self.m_lo, self.m_hi = 1 + (2 if first else 0), 1 + (1 if last else 0)   # planes I get from below / above

        def pack(k, cond):
            return torch.zeros((k, *plane), dtype=f, device=dev) if cond else None

        self.send_lo, self.send_hi = pack(self.k_lo, self.nb_lo is not None), pack(self.k_hi, self.nb_hi is not None)
        self.recv_lo, self.recv_hi = pack(self.m_lo, self.nb_lo is not None), pack(self.m_hi, self.nb_hi is not None)

        def pl(t, i, cond=True):
            return t[i] if (t is not None and cond) else None

        self.bufs = {
            "sums": self.sums,
            "r_send_lo": pl(self.send_lo, 0), "r_send_hi": pl(self.send_hi, 0),
            "r_recv_lo": pl(self.recv_lo, 0), "r_recv_hi": pl(self.recv_hi, 0),
            "x_ghost_lo": buf(self.nb_lo is not None), "x_ghost_hi": buf(self.nb_hi is not None),
            "bc_far_lo0": pl(self.recv_lo, 1, first), "bc_far_lo1": pl(self.recv_lo, 2, first),
            "bc_far_hi0": pl(self.recv_hi, 1, last),
            "x_pack_lo1": pl(self.send_lo, 1, first), "x_pack_hi0": pl(self.send_hi, 1, last),
            "x_pack_hi1": pl(self.send_hi, 2, last),
        }
        self._iter_ops = None
        self.terms = list(terms)
        self._stage = None  # pinned CPU staging when the process group cannot move GPU tensors
        self.lib_comm = self._setup_lib_comm() if self.uses_lib_comm else False
        self.folded = False

    # -- RCCL inside the library ---------------------------------------------------
    def _setup_lib_comm(self) -> bool:
        """Give the library its own RCCL communicator so that ``iterate(n)`` is ONE C call (kernels,
        all-reduces and the plane exchange on one stream, no Python between the phases).  Used when the
        process group is RCCL and every rank passes the library's collective self-test; otherwise the
        stepwise torch.distributed path below stays (PYAPES_HIP_COMM=0 forces that)."""
        be, d = self.be, self.dist
        if not hasattr(be, "comm_init") or not be.get_option("comm"):     # option "comm" / PYAPES_HIP_COMM=0
            return False
        # a stand-in for librccl handed to the library by an explicit pa_comm_use_impl() call (tests: ranks as processes
        # that may share one GPU, tests/lib): the process group is gloo then and the small agreement tensors live on the host
        standin = be.comm_impl().startswith("custom")
        if not self.x.is_cuda or (d.get_backend(self.group) != "nccl" and not standin):
            return False
        if getattr(be, "comm_ready", None) == (self.rank, self.world):   # an earlier solve on this mesh made it
            be.comm_plan(self.nb_lo, self.nb_hi, self.send_lo, self.send_hi, self.recv_lo, self.recv_hi)
            return True
        dev = self.x.device if d.get_backend(self.group) == "nccl" else torch.device("cpu")
        self._agree_dev = dev
        # every step that can fail on ONE rank is followed by an agreement (all-reduce MIN of an ok flag)
        # before any rank enters a call the others must join: ncclCommInitRank blocks until all ranks
        # arrive, so a rank that could not even load librccl must make everybody skip it
        ok = 1 if (not hasattr(be, "comm_available") or be.comm_available()) else 0
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        try:
            if ok and self.rank == 0:
                uid.copy_(torch.frombuffer(bytearray(be.comm_unique_id()), dtype=torch.uint8))
        except Exception:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        src = d.get_global_rank(self.group, 0) if self.group is not None else 0
        d.all_reduce(flag, op=d.ReduceOp.MIN, group=self.group)
        if int(flag.item()) == 0:
            return False
        d.broadcast(uid, src=src, group=self.group)
        self.lib_comm_error = None
        try:
            be.comm_init(self.rank, self.world, bytes(uid.cpu().numpy().tobytes()))
            be.comm_selftest(float(min(be.get_option("comm_timeout"), 30)))
        except Exception as e:   # this rank is out; the agreement below takes every rank to the stepwise driver
            ok = 0
            self.lib_comm_error = str(e)
        flag.fill_(ok)
        d.all_reduce(flag, op=d.ReduceOp.MIN, group=self.group)
        if int(flag.item()) == 0:
            try:
                be.comm_destroy()
            except Exception:
                pass
            return False
        be.comm_plan(self.nb_lo, self.nb_hi, self.send_lo, self.send_hi, self.recv_lo, self.recv_hi)
        be.comm_ready = (self.rank, self.world)
        return True

    # -- communication -----------------------------------------------------------
    def _p2p(self, sends: list[tuple[Tensor, int, int]], recvs: list[tuple[Tensor, int, int]]) -> None:
        """sends/recvs: (tensor, peer, tag).  One batched, ordered exchange."""
        if not sends and not recvs:
            return
        d = self.dist
        gpu = bool(sends and sends[0][0].is_cuda) or bool(recvs and recvs[0][0].is_cuda)
        staged = gpu and d.get_backend(self.group) != "nccl"
        if staged:  # gloo with GPU tensors (2-rank rehearsal on one card): go through the host
            torch.cuda.synchronize()
            s2 = [(t.cpu(), p, tag) for t, p, tag in sends]
            r2 = [(torch.empty(t.shape, dtype=t.dtype), p, tag) for t, p, tag in recvs]
        else:
            s2, r2 = sends, recvs
        ops = [d.P2POp(d.isend, t, p, self.group, tag) for t, p, tag in s2]
        ops += [d.P2POp(d.irecv, t, p, self.group, tag) for t, p, tag in r2]
        for w in d.batch_isend_irecv(ops):
            w.wait()
        if staged:
            for (dst, _, _), (src, _, _) in zip(recvs, r2):
                dst.copy_(src)

    def _plane_ops(self, lo_send, hi_send, lo_recv, hi_recv):
        """(sends, recvs) for: my first plane down / my last plane up, ghosts in.  Tag 0 = travelling up
        (lands in a lower ghost), tag 1 = travelling down.  Receives are listed hi first so that with
        P = 2 and a periodic ring (both neighbours are the same peer) the two messages pair correctly
        even on a backend that ignores tags (NCCL matches same-peer operations in order)."""
        sends, recvs = [], []
        if self.nb_lo is not None and lo_send is not None:
            sends.append((lo_send, self.nb_lo, 1))
        if self.nb_hi is not None and hi_send is not None:
            sends.append((hi_send, self.nb_hi, 0))
        if self.nb_hi is not None and hi_recv is not None:
            recvs.append((hi_recv, self.nb_hi, 1))
        if self.nb_lo is not None and lo_recv is not None:
            recvs.append((lo_recv, self.nb_lo, 0))
        return sends, recvs

    def _bc_far_ops(self, x: Tensor | None = None):
        """Periodic axis 0: planes of x the end ranks' BC fill reads across the ring."""
        x = self.x if x is None else x
        sends, recvs = [], []
        if self.periodic0:
            P, r = self.world, self.rank
            if r == P - 1:
                sends += [(x[-1], 0, 2), (x[-2], 0, 3)]
                recvs += [(self.bufs["bc_far_hi0"], 0, 4)]
            if r == 0:
                sends += [(x[1], P - 1, 4)]
                recvs += [(self.bufs["bc_far_lo0"], P - 1, 2), (self.bufs["bc_far_lo1"], P - 1, 3)]
        return sends, recvs

    def _exchange_planes(self, lo_send, hi_send, lo_recv, hi_recv) -> None:
        self._p2p(*self._plane_ops(lo_send, hi_send, lo_recv, hi_recv))

    def _exchange_bc_far(self) -> None:
        self._p2p(*self._bc_far_ops())

    def _exchange_iter(self) -> None:
        """The one batched exchange of an iteration: ONE packed message to each neighbour (residual
        plane + the periodic x planes phase_b packed behind it) and one from each.  Same-peer order
        (P = 2 ring): sends [down, up] against receives [from above, from below] on both sides."""
        self._p2p(*self._plane_ops(self.send_lo, self.send_hi, self.recv_lo, self.recv_hi))

    def _allreduce(self, lo: int, hi: int) -> None:
        self.dist.all_reduce(self.sums[lo:hi], op=self.dist.ReduceOp.SUM, group=self.group)


class SlabCG(_SlabDriver):
    """Stepwise CG over P slabs.  ``begin`` / ``iterate(n)`` / ``end`` mirror pa_cg_begin /
    pa_cg_iterate / pa_cg_end of the single-GPU path; nothing in ``iterate`` synchronises
    the host with the device."""

    uses_lib_comm = True

    # -- solve ------------------------------------------------------------------------
    def begin(self, tol: float, max_it: int, adjust_rhs: bool = True) -> None:
        be = self.be
        be.slab_set(self.bufs)
        be.bind_bcs(self.var(), self.var.bcs, 0)
        be.set_terms(self.terms)
        if adjust_rhs:
            be.rhs_adjust(self.rhs)
        self._exchange_bc_far()
        be.apply_bc_bound(self.x)                      # linalg.py:97, before the ghosts of x move
        self._exchange_planes(self.x[0], self.x[-1], self.bufs["x_ghost_lo"], self.bufs["x_ghost_hi"])
        be.cg_begin(self.x, self.rhs, tol, max_it)      # r, d = r, local sum r.r -> sums[1], r planes
        self._exchange_planes(self.bufs["r_send_lo"], self.bufs["r_send_hi"],
                              self.bufs["r_recv_lo"], self.bufs["r_recv_hi"])
        self._allreduce(1, 2)
        self.folded = self._agree_fold()

    def _agree_fold(self) -> bool:
        """Folded iterations of the library-side loop (include/pyapes_hip.h): every rank plans its partial-row
        counts, the ranks take the MAX -- or stay stepwise, all of them, if a single rank cannot fold."""
        be, d = self.be, self.dist
        if not self.lib_comm or not hasattr(be, "cg_fold_plan") or not be.get_option("slab_fold"):   # PYAPES_HIP_SLAB_FOLD=0
            return False
        rows = be.cg_fold_plan()
        t = torch.tensor([*rows, -min(rows[0], rows[1])], dtype=torch.int64, device=getattr(self, "_agree_dev", self.x.device))
        d.all_reduce(t, op=d.ReduceOp.MAX, group=self.group)
        agreed = [int(v) for v in t[:3].tolist()]
        if -int(t[3].item()) <= 0:          # some rank has no tiled phase kernels here
            return False
        be.cg_fold_set(agreed)
        return True

    def iterate(self, n: int) -> None:
        be = self.be
        if self.lib_comm:
            be.cg_iterate_comm(n)
            return
        for _ in range(n):
            be.cg_phase_a()                              # d' = r + beta d ; local sum d'.Ad'
            self._allreduce(0, 1)
            be.cg_phase_b()                              # alpha ; x, r update ; r planes out
            self._exchange_iter()
            be.cg_bc()                                   # BC fill of x ; local sums r.r, |dx|^2
            self._allreduce(1, 3)
            be.cg_finish_iter()                          # beta, stop test, itr (device side)

    def solve(self, tol: float, max_it: int, poll: int = 8, adjust_rhs: bool = True) -> Any:
        """Run to the reference's stop rule (tol / max_it + 1 iterations); polls the device-side
        done flag every ``poll`` iterations -- iterations enqueued after it is set are no-ops."""
        self.begin(tol, max_it, adjust_rhs=adjust_rhs)
        done = 0
        while done <= max_it:
            n = min(poll, max_it + 1 - done)
            self.iterate(n)
            done += n
            if self.be.report().itr < done:
                break
        return self.end()

    def profile(self, n: int) -> dict[str, float]:
        self.be.profile(True)
        self.iterate(n)
        out = self.be.profile_read()
        self.be.profile(False)
        return out

    def end(self) -> Any:
        rep = self.be.cg_end()
        self.be.slab_set(None)
        return rep


class SlabBiCGSTAB(_SlabDriver):
    """Stepwise BiCGSTAB over P slabs (linalg.py:162-279; include/pyapes_hip.h "stepwise BiCGSTAB on a slab").  Per
    iteration: the boundary planes of v' = A p' and of the new residual (with the periodic x planes behind them) go to
    the axis-0 neighbours, three small all-reduces carry r0.v', then (|s|^2, t.s, t.t, r0.t), then |r|^2; the direction
    p is never sent -- its ghost planes follow the owner's recurrence on every rank.  Communication through
    RCCL inside the library where the process group allows it (``pa_bicg_iterate_comm``: n iterations, one C call),
    else ``torch.distributed`` between the step calls (gloo in the rehearsals)."""

    uses_lib_comm = True

    def __init__(self, mesh: Any, var: Any, rhs: Tensor, terms: Sequence[dict], dist: Any,
                 backend: Any = None, group: Any = None):
        super().__init__(mesh, var, rhs, terms, dist, backend, group)
        dev, f, plane = self.x.device, self.x.dtype, tuple(self.x.shape[1:])

        def buf(cond):
            return torch.zeros(plane, dtype=f, device=dev) if cond else None

        self.v_send_lo, self.v_send_hi = buf(self.nb_lo is not None), buf(self.nb_hi is not None)
        self.v_recv_lo, self.v_recv_hi = buf(self.nb_lo is not None), buf(self.nb_hi is not None)

    def begin(self, tol: float, max_it: int, adjust_rhs: bool = True) -> None:
        be = self.be
        be.slab_set(self.bufs)
        be.slab_set_v(self.v_send_lo, self.v_send_hi, self.v_recv_lo, self.v_recv_hi)
        be.bind_bcs(self.var(), self.var.bcs, 0)
        be.set_terms(self.terms)
        if adjust_rhs:
            be.rhs_adjust(self.rhs)
        for t in (self.v_recv_lo, self.v_recv_hi):      # v = 0 at the start (linalg.py:191), on the ghost planes too
            if t is not None:
                t.zero_()
        self._exchange_bc_far()
        be.apply_bc_bound(self.x)                      # linalg.py:181, before the ghosts of x move
        self._exchange_planes(self.x[0], self.x[-1], self.bufs["x_ghost_lo"], self.bufs["x_ghost_hi"])
        be.bicg_begin(self.x, self.rhs, tol, max_it)    # r0 = r, local r0.r0 -> sums[1], r planes
        self._exchange_planes(self.bufs["r_send_lo"], self.bufs["r_send_hi"],
                              self.bufs["r_recv_lo"], self.bufs["r_recv_hi"])
        self._allreduce(1, 2)
        be.bicg_start()                                 # rho' = r0.r0, the first beta (device side)

    def iterate(self, n: int) -> None:
        be = self.be
        if self.lib_comm:
            be.bicg_iterate_comm(n)
            return
        for _ in range(n):
            be.bicg_pv()                                 # p', v' = A p' ; local r0.v' ; v' planes out
            self._allreduce(0, 1)
            self._exchange_planes(self.v_send_lo, self.v_send_hi, self.v_recv_lo, self.v_recv_hi)
            be.bicg_st()                                 # alpha ; s, t = A s ; local |s|^2, t.s, t.t, r0.t
            self._allreduce(1, 5)
            be.bicg_x()                                  # stop test 1, omega, rho' ; x, r update ; planes out
            self._exchange_iter()
            be.bicg_bc()                                 # BC fill of x ; local |r|^2
            self._allreduce(5, 6)
            be.bicg_finish()                             # stop test 2, beta, rho (device side)

    def solve(self, tol: float, max_it: int, poll: int = 8, adjust_rhs: bool = True) -> Any:
        """Run to the reference's stop rule (either stop test, or max_it iterations: linalg.py:262-271)."""
        self.begin(tol, max_it, adjust_rhs=adjust_rhs)
        done, limit = 0, max(int(max_it), 1)
        while done < limit:
            n = min(poll, limit - done)
            self.iterate(n)
            done += n
            if self.be.report().itr < done:
                break
        return self.end()

    def end(self) -> Any:
        rep = self.be.bicg_end()
        self.be.slab_set(None)
        return rep


class SlabJacobi(_SlabDriver):
    """Stepwise Jacobi over P slabs (the sweep of ``pa_jacobi`` -- [new, SURVEY a15] -- split at its exchanges; include/
    pyapes_hip.h "stepwise Jacobi on a slab").  Per sweep: on a periodic ring the far planes of the new iterate travel
    between the end ranks for the BC fill, one all-reduce carries |dx|^2, and the first / last owned plane of the new
    iterate lands in the neighbours' ghost planes of x.  RCCL inside the library where the process group allows it
    (``pa_jacobi_iterate_comm``), else ``torch.distributed`` between the step calls."""

    uses_lib_comm = True

    def __init__(self, mesh: Any, var: Any, rhs: Tensor, terms: Sequence[dict], dist: Any,
                 backend: Any = None, group: Any = None, omega: float = 1.0):
        super().__init__(mesh, var, rhs, terms, dist, backend, group)
        self.omega = float(omega)

    def _far_ops(self):
        """The periodic x planes packed behind the first plane of the send buffers (``x_pack_*``), without that plane."""
        def tail(t):
            return t[1:] if (t is not None and t.shape[0] > 1) else None
        return self._plane_ops(tail(self.send_lo), tail(self.send_hi), tail(self.recv_lo), tail(self.recv_hi))

    def begin(self, tol: float, max_it: int, adjust_rhs: bool = True) -> None:
        be = self.be
        be.slab_set(self.bufs)
        be.bind_bcs(self.var(), self.var.bcs, 0)
        be.set_terms(self.terms)
        if adjust_rhs:
            be.rhs_adjust(self.rhs)
        self._exchange_bc_far()
        be.apply_bc_bound(self.x)
        self._exchange_planes(self.x[0], self.x[-1], self.bufs["x_ghost_lo"], self.bufs["x_ghost_hi"])
        be.jacobi_begin(self.x, self.rhs, tol, max_it, self.omega)

    def iterate(self, n: int) -> None:
        be = self.be
        if self.lib_comm:
            be.jacobi_iterate_comm(n)
            return
        for _ in range(n):
            be.jacobi_sweep()                            # x' on S ; periodic far planes of x' out
            if self.periodic0:
                self._p2p(*self._far_ops())
            be.jacobi_bc()                               # BC fill of x' ; local |dx|^2 ; first / last plane out
            self._allreduce(2, 3)
            self._exchange_planes(self.bufs["r_send_lo"], self.bufs["r_send_hi"],
                                  self.bufs["x_ghost_lo"], self.bufs["x_ghost_hi"])
            be.jacobi_finish()                           # stop test, sweep count (device side)

    def solve(self, tol: float, max_it: int, poll: int = 8, adjust_rhs: bool = True) -> Any:
        """Run to the reference's stop rule as the CG has it (tol / max_it + 1 sweeps)."""
        self.begin(tol, max_it, adjust_rhs=adjust_rhs)
        done = 0
        while done <= max_it:
            n = min(poll, max_it + 1 - done)
            self.iterate(n)
            done += n
            if self.be.report().itr < done:
                break
        return self.end()

    def end(self) -> Any:
        rep = self.be.jacobi_end()
        self.be.slab_set(None)
        return rep


class SlabEuler(_SlabDriver):
    """The explicit Euler march (``solver/march.py``, [new, SURVEY a15]) over P slabs: per step the first / last owned
    plane of phi goes into the neighbours' ghost planes, every rank runs the step kernel on its planes
    (``pa_euler_step`` in slab mode: no fill), on a periodic axis 0 the far planes of the NEW field travel between the
    end ranks, and every rank fills its BCs.  No reduction: nothing but neighbour planes crosses the ranks."""

    def __init__(self, mesh: Any, var: Any, dist: Any, backend: Any = None, group: Any = None):
        super().__init__(mesh, var, var(), [], dist, backend, group)

    def march(self, kind: int, u: Any, nu: float, dt: float, nsteps: int) -> Tensor:
        """-> the tensor (one of the two ping-pong buffers) that holds the state after ``nsteps`` steps."""
        be = self.be
        be.slab_set(self.bufs)
        be.bind_bcs(self.var(), self.var.bcs, 0)
        cur, nxt = self.x, torch.empty_like(self.x)
        try:
            for _ in range(int(nsteps)):
                self._exchange_planes(cur[0], cur[-1], self.bufs["x_ghost_lo"], self.bufs["x_ghost_hi"])
                be.euler_step(cur, nxt, kind, u, nu, dt)
                if self.periodic0:
                    self._p2p(*self._bc_far_ops(nxt))
                be.apply_bc_bound(nxt)
                cur, nxt = nxt, cur
        finally:
            be.slab_set(None)
        return cur


def slab_solver(method: str, mesh: Any, var: Any, rhs: Tensor, terms: Sequence[dict], dist: Any = None,
                backend: Any = None, group: Any = None, omega: float = 1.0) -> Any:
    """The slab driver ``linalg.solve`` hands a solve on ``Mesh(..., slab=(rank, world))`` to."""
    if dist is None:
        import torch.distributed as dist_mod
        dist = dist_mod
        if not dist.is_initialized():
            raise RuntimeError("pyapes_amd: a solve on a slab mesh needs torch.distributed to be initialised "
                               "(one process per GPU, backend 'nccl' = RCCL)")
    if method == "cg":
        return SlabCG(mesh, var, rhs, terms, dist, backend, group)
    if method == "bicgstab":
        return SlabBiCGSTAB(mesh, var, rhs, terms, dist, backend, group)
    if method == "jacobi":
        return SlabJacobi(mesh, var, rhs, terms, dist, backend, group, omega=omega)
    raise NotImplementedError(f"pyapes_amd: no slab driver for method '{method}' (cg, bicgstab, jacobi)")
