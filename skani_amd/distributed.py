"""Multi-GPU triangle: one process per GPU.  The whole protocol lives BELOW the C ABI (csrc/dist.hip: skh_triangle_distributed -- marker
all-gather, row-sharded screen, balanced order-independent assignment of the candidate pairs, point-to-point exchange of exactly the
sketches that have to move, result all-gather); this module only makes the communicator:

  * `Comm.rccl(ctx, dist, rank, world)`   -- the product path: the library talks RCCL itself (device buffers over xGMI); torch.distributed is
    used once, to hand rank 0's RCCL unique id to the other ranks (any launcher-side channel would do: skh_comm_unique_id / skh_comm_create_rccl);
  * `Comm.host(ctx, dist, rank, world)`   -- host-memory collectives supplied by the caller, here torch.distributed on CPU tensors ("gloo").
    This is how the CPU tests drive world sizes 2..4 through the very same C++ protocol (a Rust/MPI host would pass MPI_Allgather / MPI_Alltoallv).
"""
import ctypes as C

import numpy as np

from . import _binding as B


class Comm:
    def __init__(self, ctx, handle, rank, world, keep=None):
        self.ctx, self.h, self.rank, self.world, self._keep = ctx, handle, rank, world, keep

    @staticmethod
    def rccl(ctx, dist, rank, world, torch=None, device=None):
        """RCCL communicator inside the library.  `dist` (an initialised torch.distributed) only carries the 128-byte unique id."""
        if torch is None:
            import torch as _t
            torch = _t
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            ctx.check(ctx.L.skh_comm_unique_id(ident))
        if world > 1:
            dev = device if (device is not None and "nccl" in str(dist.get_backend())) else torch.device("cpu")
            t = torch.tensor(list(bytes(ident)), dtype=torch.uint8, device=dev)
            dist.broadcast(t, src=0)
            ident = (C.c_uint8 * 128)(*[int(x) for x in t.cpu().tolist()])
        h = C.c_void_p()
        ctx.check(ctx.L.skh_comm_create_rccl(ctx.h, ident, rank, world, C.byref(h)))
        return Comm(ctx, h, rank, world)

    @staticmethod
    def host(ctx, dist, rank, world, torch=None):
        """Host-memory collectives on torch.distributed CPU tensors (gloo): every buffer the library hands over is host memory."""
        if torch is None:
            import torch as _t
            torch = _t

        def view(ptr, n):
            return torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * max(int(n), 1)).from_address(ptr))[:int(n)]) if n else torch.zeros(0, dtype=torch.uint8)

        def all_gather(_user, send, recv, nbytes):
            try:
                if nbytes == 0:
                    return 0
                s = view(send, nbytes); r = view(recv, nbytes * world)
                if world == 1:
                    r.copy_(s)
                else:
                    dist.all_gather(list(r.chunk(world)), s)
                return 0
            except Exception as e:      # never unwind into C
                print("skani_amd.distributed: all_gather failed:", repr(e), flush=True)
                return 1

        def all_to_all_v(_user, send, scnt, soff, recv, rcnt, roff):
            try:
                sc = [int(scnt[r]) for r in range(world)]; so = [int(soff[r]) for r in range(world)]
                rc = [int(rcnt[r]) for r in range(world)]; ro = [int(roff[r]) for r in range(world)]
                # the library lays both buffers out in rank order without gaps, which is what all_to_all_single wants
                assert all(so[r] == sum(sc[:r]) for r in range(world)) and all(ro[r] == sum(rc[:r]) for r in range(world))
                s = view(send, sum(sc)); r = view(recv, sum(rc))
                if world == 1:
                    r.copy_(s)
                else:
                    dist.all_to_all_single(r, s, output_split_sizes=rc, input_split_sizes=sc)
                return 0
            except Exception as e:
                print("skani_amd.distributed: all_to_all_v failed:", repr(e), flush=True)
                return 1

        hc = B.HostCollectives(None, B.ALL_GATHER_FN(all_gather), B.ALL_TO_ALL_V_FN(all_to_all_v))
        h = C.c_void_p()
        ctx.check(ctx.L.skh_comm_create_host(ctx.h, C.byref(hc), rank, world, C.byref(h)))
        return Comm(ctx, h, rank, world, keep=hc)        # the callbacks must outlive the communicator

    def triangle(self, ss_local, map_params, identity=0.0, rescue_small=True, rows_to_root=False):
        """Collective.  Returns (i, j, results, n_chained_total, stats) with global genome indices: the whole triangle on every rank, or -- rows_to_root
        (SKH_DIST_ROWS_TO_ROOT) -- on rank 0 only, the other ranks getting the rows of the pairs they chained."""
        L = self.ctx.L
        oi, oj, orr, n, nch = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        st = B.DistStats()
        self.ctx.check(L.skh_triangle_distributed_ex(self.ctx.h, self.h, ss_local.h, identity, int(rescue_small), C.byref(map_params), 1 if rows_to_root else 0,
                                                     C.byref(oi), C.byref(oj), C.byref(orr), C.byref(n), C.byref(nch), C.byref(st)))
        try:
            from .api import _take_rows
            i, j, res = _take_rows(n.value, oi, oj, orr)
        finally:
            L.skh_free(oi); L.skh_free(oj); L.skh_free(orr)
        return i, j, res, nch.value, {nm: getattr(st, nm) for nm, _ in st._fields_}

    def selftest(self):
        """Collective: small gathers and exchanges through the communicator, every byte checked (skh_comm_selftest)."""
        self.ctx.check(self.ctx.L.skh_comm_selftest(self.ctx.h, self.h))

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.skh_comm_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def distributed_triangle(ctx, ss_local, params, map_params, dist, rank, world, identity=0.0, rescue_small=True, torch=None, device=None, comm=None,
                         with_stats=False, rows_to_root=False):
    """ss_local: this rank's sketches (any number, also none), created with genome_rank = the genome's rank in an ordering common to all
    ranks.  Global genome index = genomes on lower ranks + local index.  Returns (i, j, results, n_chained_total) on every rank.
    Without `comm` a host-collective communicator over `dist` is made for the call (tests); bench.py passes an RCCL one."""
    if world == 1 and comm is None:
        r = ctx.triangle(ss_local, map_params, identity, rescue_small)
        return (r + ({},)) if with_stats else r
    own = comm is None
    if own:
        comm = Comm.host(ctx, dist, rank, world, torch=torch)
    try:
        i, j, res, n, st = comm.triangle(ss_local, map_params, identity, rescue_small, rows_to_root=rows_to_root)
    finally:
        if own:
            comm.close()
    return (i, j, res, n, st) if with_stats else (i, j, res, n)


def plan_pairs(lib, n_genomes, pair_i, pair_j, weight, world, holder=None):
    """skh_plan_pairs: the rank every candidate pair would be chained on (host only; `lib` = a loaded library, e.g. Context.L)."""
    pi = np.ascontiguousarray(pair_i, np.uint32); pj = np.ascontiguousarray(pair_j, np.uint32); w = np.ascontiguousarray(weight, np.uint64)
    owner = np.zeros(len(pi), np.uint8)
    h = np.ascontiguousarray(holder, np.uint32) if holder is not None else None
    rc = lib.skh_plan_pairs(n_genomes, pi.ctypes.data_as(C.c_void_p), pj.ctypes.data_as(C.c_void_p), len(pi), w.ctypes.data_as(C.c_void_p),
                            h.ctypes.data_as(C.c_void_p) if h is not None else None, world, owner.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError("skh_plan_pairs failed with %d" % rc)
    return owner
