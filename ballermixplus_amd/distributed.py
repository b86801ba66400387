"""Multi-GPU: shard test sites over one process per GPU, gather the result records once.

The path has no exchange step: every output row depends only on the
read-only site arrays around its test site, so each rank holds the full site arrays of
the chromosome it works on plus the (<1 MB) table, scans its share of the test sites,
and the 16-byte result records (CLR f64, linear grid index i32, nSites i32) are gathered
to every rank with one all_gather per array -- RCCL over xGMI when the backend is 'nccl',
gloo on CPU in the tests.  Results do not depend on the number of ranks: a window is
always reduced by one workgroup set in the same order.
"""
import os

import numpy as np

BLOCK = 4096   # test sites are dealt to ranks in blocks of this many (balances density changes)


def assign(M, world, block=BLOCK):
    """Index arrays, one per rank: blocks of `block` consecutive test sites dealt round-robin."""
    nblk = (M + block - 1) // block
    out = []
    for r in range(world):
        parts = [np.arange(b * block, min((b + 1) * block, M), dtype=np.int64) for b in range(r, nblk, world)]
        out.append(np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64))
    return out


class _DevArray:
    """Zero-copy view of a device buffer owned by libbmxscan for torch.as_tensor."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': typestr, 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}


class World:
    def __init__(self, rank=0, size=1, local_rank=0, backend=None):
        self.rank, self.size, self.local_rank, self.backend = rank, size, local_rank, backend
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
                torch.cuda.set_device(local)
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

    # ------------------------------------------------------------------ gather
    def all_gather_records(self, clr, lin, ns, counts):
        """Gather per-rank (clr f64, lin i32, ns i32) arrays of lengths `counts` to every rank.
        Inputs are numpy arrays (gloo) or torch CUDA tensors (nccl); returns numpy arrays per rank."""
        import torch
        import torch.distributed as dist
        pad = max(counts) if counts else 0
        dev = torch.device('cuda', self.local_rank) if self.backend == 'nccl' else torch.device('cpu')

        def prep(a, dtype):
            t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
            t = t.to(device=dev, dtype=dtype)
            if t.numel() < pad:
                t = torch.cat([t, torch.zeros(pad - t.numel(), dtype=dtype, device=dev)])
            return t.contiguous()

        outs = []
        for a, dtype in ((clr, torch.float64), (lin, torch.int32), (ns, torch.int32)):
            mine = prep(a, dtype)
            buf = torch.empty(pad * self.size, dtype=dtype, device=dev)
            dist.all_gather_into_tensor(buf, mine)
            outs.append(buf.cpu().numpy().reshape(self.size, pad))
        return [[o[r, :counts[r]] for r in range(self.size)] for o in outs]

    # ------------------------------------------------------------------ runner
    def sharded_runner(self, compute=None):
        """A drop-in for engine.scan_batch that scans only this rank's test sites and
        all-gathers the records.  `compute(sel, test_gen, lo, hi) -> (clr, lin, ns)` defaults to
        the GPU scan; tests inject a CPU function to exercise the sharding on gloo."""
        world = self

        def run(sel, test_gen, win_lo, win_hi):
            test_gen = np.asarray(test_gen, dtype=np.float64)
            win_lo = np.asarray(win_lo, dtype=np.int64)
            win_hi = np.asarray(win_hi, dtype=np.int64)
            M = len(test_gen)
            parts = assign(M, world.size)
            mine = parts[world.rank]
            counts = [len(p) for p in parts]
            if compute is not None:
                clr, lin, ns = compute(sel, test_gen[mine], win_lo[mine], win_hi[mine])
            elif len(mine):
                import torch
                sel.ctx.set_tests(test_gen[mine], win_lo[mine], win_hi[mine])
                sel.ctx.scan()
                sel.ctx.sync()
                if world.backend == 'nccl':
                    pc, pl, pn = sel.ctx.result_ptrs()
                    dev = torch.device('cuda', world.local_rank)
                    clr = torch.as_tensor(_DevArray(pc, len(mine), '<f8'), device=dev)
                    lin = torch.as_tensor(_DevArray(pl, len(mine), '<i4'), device=dev)
                    ns = torch.as_tensor(_DevArray(pn, len(mine), '<i4'), device=dev)
                else:
                    c, ix, ia, iA, n = sel.ctx.fetch()
                    npairs = len(sel.grid_x) * len(sel.grid_abeta)
                    clr, ns = c, n
                    lin = np.where(iA < 0, -1, iA * npairs + ix * len(sel.grid_abeta) + ia).astype(np.int32)
            else:
                clr, lin, ns = np.zeros(0), np.zeros(0, np.int32), np.zeros(0, np.int32)
            g_clr, g_lin, g_ns = world.all_gather_records(clr, lin, ns, counts)
            clr_all = np.empty(M, dtype=np.float64)
            lin_all = np.empty(M, dtype=np.int32)
            ns_all = np.empty(M, dtype=np.int32)
            for r in range(world.size):
                clr_all[parts[r]] = g_clr[r]
                lin_all[parts[r]] = g_lin[r]
                ns_all[parts[r]] = g_ns[r]
            return unpack_lin(clr_all, lin_all, ns_all, len(sel.grid_x), len(sel.grid_abeta))

        return run


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
