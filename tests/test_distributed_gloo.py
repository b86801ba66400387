"""The N>1 path on CPU: world_size 2 over gloo.  The GPU scan is replaced by the C oracle as the
per-rank compute function; what is under test is the sharding (blocks of 4096 test sites dealt
round-robin), the all_gather of the (CLR, linear index, nSites) records and the reassembly."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import cases
    from util import c_oracle, c_scan
    from ballermixplus_amd import distributed
    w = distributed.World.from_env(backend='gloo')
    argv, gold = cases.ALL_CASES['ex1_B2']
    opt, case, ts = cases.host_side(argv)
    m = case.oracle_model()
    L = c_oracle()
    nx, nab = len(case.xs), len(case.abetas)

    def compute(sel, tg, lo, hi):
        clr, ix, ia, iA, ns = c_scan(L, m.R, case.As, case.data.genPos, m.row, tg, lo, hi)
        lin = np.where(iA < 0, -1, (iA * nx + ix) * nab + ia).astype(np.int32)
        return clr, lin, ns

    class Sel:
        grid_x, grid_abeta = case.xs, case.abetas

    distributed.BLOCK = 64            # 757 test sites -> 12 blocks, dealt 6/6
    run = w.sharded_runner(compute=compute)
    res = run(Sel, ts.test_gen, ts.lo, ts.hi)
    if rank == 0:
        q.put([np.asarray(a) for a in res])
    w.finish()


def test_two_rank_sharded_scan_equals_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, HERE)
    import cases
    from util import c_oracle, c_scan
    argv, gold = cases.ALL_CASES['ex1_B2']
    opt, case, ts = cases.host_side(argv)
    m = case.oracle_model()
    ref = c_scan(c_oracle(), m.R, case.As, case.data.genPos, m.row, ts.test_gen, ts.lo, ts.hi)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)          # bitwise: sharding must not change any row


def test_assignment_covers_every_test_site_once():
    from ballermixplus_amd import distributed
    for M, W in [(1, 1), (757, 2), (4096 * 3 + 5, 4), (100000, 8), (10, 8)]:
        parts = distributed.assign(M, W)
        allidx = np.concatenate(parts)
        assert len(allidx) == M and np.array_equal(np.sort(allidx), np.arange(M))
        for p in parts:          # every shard starts on a multiple of the kernel's group size
            assert len(p) == 0 or all(int(b) % 16 == 0 for b in p[::distributed.BLOCK][:4])


def test_unpack_lin_roundtrip():
    from ballermixplus_amd import distributed
    lin = np.array([-1, 0, 509, 510, 15809], dtype=np.int32)
    clr, ix, ia, iA, ns = distributed.unpack_lin(np.zeros(5), lin, np.zeros(5, np.int32), 10, 51)
    assert iA.tolist() == [-1, 0, 0, 1, 30] and ix.tolist() == [-1, 0, 9, 0, 9] and ia.tolist() == [-1, 0, 50, 0, 50]
