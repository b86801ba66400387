"""Multi-GPU: shard test sites over one process per GPU, gather the result records once.

The path has no exchange step: every output row depends only on the
read-only site arrays around its test site, so each rank holds the full site arrays of
the chromosome it works on plus the (<1 MB) table, scans its share of the test sites,
and the 16-byte result records (bmx_record: CLR f64, linear grid index i32, nSites i32) go to
the rank that writes the output with ONE gather -- RCCL over xGMI when the backend is 'nccl'
(every peer has its own link to rank 0), gloo on CPU in the tests.  Results do not depend on
the number of ranks: a window is always reduced by one workgroup set in the same order.
"""
import os

import numpy as np

BLOCK = 4096   # test sites are dealt to ranks in blocks of this many (balances density changes; a multiple of the
               # kernel's group size, so no window's arithmetic depends on the number of ranks)
RECORD = np.dtype([('clr', '<f8'), ('lin', '<i4'), ('nsites', '<i4')])     # bmx_record, include/bmxscan.h


def assign(M, world, block=None, weights=None):
    """Index arrays, one per rank: blocks of `block` (default: BLOCK, read at call time) consecutive test sites, dealt
    round-robin -- or, with `weights` (one estimate of the work per block: block_work), so that the ranks' total work is as
    even as it gets: heaviest block first, each to the rank with the least work so far (ties: the lower rank), every
    rank scanning its blocks in ascending order.  Blocks start on the same test sites either way, so no window's
    arithmetic depends on the choice ("chunks balanced by estimated work")."""
    block = BLOCK if block is None else int(block)
    nblk = (M + block - 1) // block
    owner = np.arange(nblk) % max(world, 1)
    if weights is not None and world > 1:
        w = np.asarray(weights, dtype=np.float64)
        if len(w) != nblk:
            raise ValueError('one weight per block is needed (%d blocks, %d weights)' % (nblk, len(w)))
        load = np.zeros(world)
        for b in np.argsort(-w, kind='stable'):
            r = int(np.argmin(load))
            owner[b] = r
            load[r] += w[b]
    out = []
    for r in range(world):
        parts = [np.arange(b * block, min((b + 1) * block, M), dtype=np.int64) for b in np.flatnonzero(owner == r)]
        out.append(np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64))
    return out


def block_work(genpos, As, zcut, test_gen, block=None):
    """Estimated work of every block of `block` test sites: block length x sum over A of the window size W_A (sites with
    A |g - t| <= zcut) at the block's middle test site -- what the scan's cost is proportional to, from two
    binary searches per A.  The density of sites per genetic unit varies along a chromosome with the recombination map."""
    block = BLOCK if block is None else int(block)
    g = np.asarray(genpos, dtype=np.float64)
    t = np.asarray(test_gen, dtype=np.float64)
    M = len(t)
    nblk = (M + block - 1) // block
    starts = np.arange(nblk, dtype=np.int64) * block
    lens = np.minimum(starts + block, M) - starts
    mid = t[starts + lens // 2]
    w = np.zeros(nblk)
    for A in As:
        r = float(zcut) / float(A)
        w += np.searchsorted(g, mid + r, 'right') - np.searchsorted(g, mid - r, 'left')
    return w * lens


class _DevArray:
    """Zero-copy view of a device buffer owned by libbmxscan for torch.as_tensor."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': typestr, 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}


class World:
    def __init__(self, rank=0, size=1, local_rank=0, backend=None):
        self.rank, self.size, self.local_rank, self.backend = rank, size, local_rank, backend
        # the GPU this rank computes on: LOCAL_RANK, or 0 for every rank when several ranks rehearse on ONE GPU
        self.device_index = 0 if os.environ.get('BMX_SINGLE_DEVICE') == '1' else local_rank
        self._own_pg = False
        self.force = False

    @classmethod
    def from_env(cls, backend=None):
        size = int(os.environ.get('WORLD_SIZE', '1'))
        rank = int(os.environ.get('RANK', '0'))
        local = int(os.environ.get('LOCAL_RANK', '0'))
        w = cls(rank, size, local, backend)
        # BMX_FORCE_DIST=1: initialise the process group even for a single rank, so that the RCCL
        # gather path can be exercised on a 1-GPU box (tests)
        w.force = os.environ.get('BMX_FORCE_DIST') == '1'
        if size > 1 or w.force:
            import torch
            import torch.distributed as dist
            if backend is None:
                backend = 'nccl' if torch.cuda.is_available() else 'gloo'
            w.backend = backend
            if backend == 'nccl':
                torch.cuda.set_device(w.device_index)
            if not dist.is_initialized():
                os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
                os.environ.setdefault('MASTER_PORT', '29533')
                dist.init_process_group(backend=backend, rank=rank, world_size=size)
                w._own_pg = True
        return w

    @property
    def distributed(self):
        return self.size > 1 or self.force

    def finish(self):
        if self.distributed:
            import torch.distributed as dist
            dist.barrier()
            if self._own_pg:
                dist.destroy_process_group()

    # ------------------------------------------------------------------ the library's own RCCL gather
    def native_comm(self, ctx):
        """engine.Comm over this world's ranks for `ctx`: rank 0 makes the RCCL id, the process group (any backend) carries its
        128 bytes to the others, every rank joins.  After this the gather needs no torch: bmx_comm_gather_records moves the packed
        records with ncclSend / ncclRecv on the context's own stream."""
        from . import engine
        box = [engine.Comm.make_id() if self.rank == 0 else None]
        if self.distributed:
            import torch.distributed as dist
            dist.broadcast_object_list(box, src=0)
        return engine.Comm(ctx, box[0], self.rank, self.size)

    # ------------------------------------------------------------------ gather
    def gather_records(self, rec, counts, dst=0):
        """ONE gather of 16-byte records to rank `dst`.  `rec`: this rank's records, either a numpy structured
        array (RECORD) or a torch int64 tensor of shape [n, 2] on the rank's GPU (nccl backend; it must not be
        rewritten before this call returns: the call waits for the collective).  Returns a list of RECORD arrays, one
        per rank, on `dst`; None elsewhere."""
        import torch
        import torch.distributed as dist
        pad = max(counts) if counts else 0
        dev = torch.device('cuda', self.device_index) if self.backend == 'nccl' else torch.device('cpu')
        if isinstance(rec, torch.Tensor):
            t = rec.reshape(-1, 2)
        else:
            t = torch.from_numpy(np.ascontiguousarray(rec, dtype=RECORD).view(np.int64).reshape(-1, 2))
        t = t.to(device=dev)
        if t.shape[0] < pad:
            t = torch.cat([t, torch.zeros((pad - t.shape[0], 2), dtype=torch.int64, device=dev)])
        t = t.contiguous()
        big = torch.empty((self.size, pad, 2), dtype=torch.int64, device=dev) if self.rank == dst else None
        dist.gather(t, list(big.unbind(0)) if self.rank == dst else None, dst=dst)
        if self.backend == 'nccl':
            # the collective runs on torch's NCCL stream: nothing may touch `rec` (the library's buffers, on the library's
            # stream) until it has finished
            torch.cuda.current_stream().synchronize()
        if self.rank != dst:
            return None
        got = big.cpu().numpy().view(RECORD).reshape(self.size, pad)
        return [got[r, :counts[r]] for r in range(self.size)]

    # ------------------------------------------------------------------ runner
    def sharded_runner(self, compute=None, block=None, balance=False):
        """A drop-in for engine.scan_batch that scans only this rank's test sites and gathers the records on
        rank 0 with ONE gather; the other ranks get None back (they write nothing).  Rank 0 gets a GatheredRecords: the
        per-rank record arrays as they arrived, which the native writer turns into rows directly (no reassembly in Python).
        `compute(sel, test_gen, lo, hi) -> (clr, lin, ns)` defaults to the GPU scan; tests inject a CPU function to
        exercise the sharding on gloo.  balance=True (the CLI's BMX_SHARD_BALANCE=1): blocks go to the ranks by estimated work
        (block_work on `sel.site_gen` / `sel.grid_A`) instead of round-robin -- same blocks, same rows, bitwise."""
        world = self

        def run(sel, test_gen, win_lo, win_hi):
            test_gen = np.asarray(test_gen, dtype=np.float64)
            win_lo = np.asarray(win_lo, dtype=np.int64)
            win_hi = np.asarray(win_hi, dtype=np.int64)
            M = len(test_gen)
            blk = BLOCK if block is None else int(block)
            weights = None
            if balance and world.size > 1:
                from . import _lib
                weights = block_work(sel.site_gen, sel.grid_A, _lib.lib().bmx_alpha_cut(), test_gen, blk)
            parts = assign(M, world.size, blk, weights)
            mine = parts[world.rank]
            counts = [len(p) for p in parts]
            if compute is not None:
                clr, lin, ns = compute(sel, test_gen[mine], win_lo[mine], win_hi[mine])
                rec = np.empty(len(mine), dtype=RECORD)
                rec['clr'], rec['lin'], rec['nsites'] = clr, lin, ns
            elif len(mine):
                sel.ctx.set_tests(test_gen[mine], win_lo[mine], win_hi[mine])
                sel.ctx.scan()
                if world.backend == 'nccl':
                    import torch
                    rec = torch.empty((len(mine), 2), dtype=torch.int64, device=torch.device('cuda', world.device_index))
                    # the selected slot's records only (a reused context may hold other chromosomes' results in other slots):
                    # waits for the scan, device -> device
                    sel.ctx.copy_records(device_ptr=rec.data_ptr(), cap=len(mine))
                else:
                    rec = sel.ctx.fetch_records()
            else:
                rec = np.zeros(0, dtype=RECORD)
            got = world.gather_records(rec, counts)
            if got is None:
                return None
            return GatheredRecords(got, M, blk, world.size, len(sel.grid_x), len(sel.grid_abeta), parts if weights is not None else None)

        return run


class GatheredRecords:
    """What rank 0 holds after the gather of a sharded scan: per_rank[r] = RECORD array of rank r's test sites in its own
    order (blocks of `block` test sites dealt round-robin)."""

    def __init__(self, per_rank, M, block, world, nx, nab, parts=None):
        self.per_rank, self.M, self.block, self.world, self.nx, self.nab = per_rank, M, block, world, nx, nab
        self.parts = parts            # the ranks' test-site indices when they are not the round-robin deal (work-balanced runs)

    def in_order(self):
        """One RECORD array in test-site order."""
        out = np.empty(self.M, dtype=RECORD)
        for r, p in enumerate(self.parts if self.parts is not None else assign(self.M, self.world, self.block)):
            out[p] = self.per_rank[r]
        return out

    def unpack(self):
        """(clr, ix, ia, iA, nsites) as engine.scan_batch returns them."""
        rec = self.in_order()
        return unpack_lin(rec['clr'].copy(), rec['lin'], rec['nsites'].copy(), self.nx, self.nab)

    def write(self, path, phys, gen, xs, abs_, As):
        """Append the rows to `path` through the native writer, straight from the per-rank arrays."""
        from . import _lib
        if self.parts is not None:       # the native writer's rank order is the round-robin deal: reassemble first
            _lib.write_records(path, phys, gen, [self.in_order()], self.block, xs, abs_, As)
        else:
            _lib.write_records(path, phys, gen, self.per_rank, self.block, xs, abs_, As)


def unpack_lin(clr, lin, ns, nx, nab):
    """linear index (iA*nx + ix)*nab + ia  ->  (ix, ia, iA); -1 stays -1."""
    lin = np.asarray(lin, dtype=np.int64)
    none = lin < 0
    npairs = nx * nab
    iA = np.where(none, -1, lin // npairs).astype(np.int32)
    p = lin % npairs
    ix = np.where(none, -1, p // nab).astype(np.int32)
    ia = np.where(none, -1, p % nab).astype(np.int32)
    return clr, ix, ia, iA, np.asarray(ns, dtype=np.int32)
