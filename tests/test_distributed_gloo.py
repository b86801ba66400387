"""The N>1 path on CPU: world_size 2 (and 3) over gloo.  The GPU scan is replaced by the C oracle as the
per-rank compute function; what is under test is the sharding (blocks of test sites dealt round-robin,
with a partial last block), the single gather of the 16-byte (CLR, linear index, nSites) records to rank 0
and the reassembly."""
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

    # 757 test sites in blocks of 64 -> 11 full blocks + one of 53, dealt round-robin: every rank scans something
    parts = distributed.assign(len(ts.test_gen), world, 64)
    assert all(len(p) > 0 for p in parts) and sum(len(p) for p in parts) == 757 and len(parts[11 % world]) % 64 == 53
    seen = []

    def counting(sel, tg, lo, hi):
        seen.append(len(tg))
        return compute(sel, tg, lo, hi)

    run = w.sharded_runner(compute=counting, block=64)
    res = run(Sel, ts.test_gen, ts.lo, ts.hi)
    assert seen == [len(parts[rank])]
    if rank == 0:
        # rank 0 holds the per-rank record arrays as they arrived; rows come straight from them through the native writer
        # (bmx_write_records), byte-identical to the rows of the reassembled arrays
        import tempfile
        from ballermixplus_amd import _lib
        assert len(res.per_rank) == world and [len(a) for a in res.per_rank] == [len(p) for p in parts]
        un = res.unpack()
        sx, sab, sA = [repr(v) for v in case.xs], [repr(v) for v in case.abetas], [repr(v) for v in case.As]
        phys = np.asarray(ts.phys, dtype=np.int64)
        genl = np.asarray(ts.gen_label, dtype=np.float64)
        with tempfile.TemporaryDirectory() as d:
            a, b = os.path.join(d, 'a.txt'), os.path.join(d, 'b.txt')
            res.write(a, phys, genl, sx, sab, sA)
            _lib.write_rows(b, phys, genl, *un, sx, sab, sA)
            assert open(a, 'rb').read() == open(b, 'rb').read() and os.path.getsize(a) > 30000
        q.put([np.asarray(a) for a in un])
    else:
        assert res is None                 # only the writing rank holds the gathered rows
    w.finish()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_scan_equals_single_process(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
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
    # the block size is read when assign() is CALLED (a default argument would have frozen 4096 at import)
    old = distributed.BLOCK
    try:
        distributed.BLOCK = 64
        assert [len(p) for p in distributed.assign(757, 2)] == [6 * 64, 5 * 64 + 53]
    finally:
        distributed.BLOCK = old
    assert [len(p) for p in distributed.assign(757, 2)] == [757, 0]


def test_record_gather_layout():
    """The 16-byte record is (f64, i32, i32) in that order: the int64-pair view used on the wire round-trips."""
    from ballermixplus_amd import _lib, distributed
    assert distributed.RECORD.itemsize == 16 and distributed.RECORD == _lib.RECORD_DTYPE
    import ctypes as C
    assert C.sizeof(_lib.BmxRecord) == 16
    rec = np.zeros(3, dtype=distributed.RECORD)
    rec['clr'], rec['lin'], rec['nsites'] = [1.5, 0.0, 7.25], [3, -1, 15809], [10, 0, 5000]
    back = rec.view(np.int64).reshape(-1, 2).copy().view(distributed.RECORD).reshape(-1)
    assert np.array_equal(back, rec)


def test_unpack_lin_roundtrip():
    from ballermixplus_amd import distributed
    lin = np.array([-1, 0, 509, 510, 15809], dtype=np.int32)
    clr, ix, ia, iA, ns = distributed.unpack_lin(np.zeros(5), lin, np.zeros(5, np.int32), 10, 51)
    assert iA.tolist() == [-1, 0, 0, 1, 30] and ix.tolist() == [-1, 0, 9, 0, 9] and ia.tolist() == [-1, 0, 50, 0, 50]


# ------------------------------------------------------------------------------------------------ work-balanced blocks
def _warped_case():
    """Example 1 with the second half of the chromosome five times denser per genetic unit (a cold region of the recombination
    map): windows there hold five times the sites, so equal-count blocks are unequal work."""
    sys.path.insert(0, HERE)
    import cases
    argv, gold = cases.ALL_CASES['ex1_B2']
    opt, case, ts = cases.host_side(argv)
    g = np.asarray(case.data.genPos, dtype=np.float64)
    h = len(g) // 2
    gen = np.where(np.arange(len(g)) < h, g, g[h] + (g - g[h]) * 0.2)
    assert np.all(np.diff(gen) >= 0)
    return case, gen


def _balanced_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    from util import c_oracle, c_scan
    from ballermixplus_amd import distributed
    w = distributed.World.from_env(backend='gloo')
    case, gen = _warped_case()
    m = case.oracle_model()
    L = c_oracle()
    nx, nab = len(case.xs), len(case.abetas)
    N = len(gen)

    def compute(sel, tg, lo, hi):
        clr, ix, ia, iA, ns = c_scan(L, m.R, case.As, gen, m.row, tg, lo, hi)
        return clr, np.where(iA < 0, -1, (iA * nx + ix) * nab + ia).astype(np.int32), ns

    class Sel:
        grid_x, grid_abeta, grid_A, site_gen = case.xs, case.abetas, case.As, gen

    tg = gen[::3]                              # every third site a test site: 253 windows in blocks of 16
    M = len(tg)
    lo, hi = np.zeros(M, np.int64), np.full(M, N - 1, np.int64)
    out = {}
    for balance in (False, True):
        res = w.sharded_runner(compute=compute, block=16, balance=balance)(Sel, tg, lo, hi)
        if rank == 0:
            assert (res.parts is not None) == balance
            out[balance] = [np.asarray(a) for a in res.unpack()]
            import tempfile
            with tempfile.TemporaryDirectory() as d:      # the writer takes either layout
                f = os.path.join(d, 'rows.txt')
                res.write(f, np.arange(M, dtype=np.int64), tg, [repr(v) for v in case.xs], [repr(v) for v in case.abetas], [repr(v) for v in case.As])
                out[('rows', balance)] = open(f, 'rb').read()
    if rank == 0:
        q.put(out)
    w.finish()


def test_work_balanced_blocks_change_no_row():
    """blocks dealt by estimated work (sum_A W_A) instead of round-robin.  Two ranks over gloo on a chromosome whose
    second half is five times denser: the same rows bitwise, from either assignment, as one process; and the heavier rank's
    share of the work drops."""
    import torch.multiprocessing as mp
    from ballermixplus_amd import _lib, distributed
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_balanced_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, HERE)
    from util import c_oracle, c_scan
    case, gen = _warped_case()
    m = case.oracle_model()
    N = len(gen)
    tg = gen[::3]
    M = len(tg)
    ref = c_scan(c_oracle(), m.R, case.As, gen, m.row, tg, np.zeros(M, np.int64), np.full(M, N - 1, np.int64))
    for balance in (False, True):
        for a, b in zip(got[balance], ref):
            assert np.array_equal(a, b)
    assert got[('rows', False)] == got[('rows', True)] and len(got[('rows', True)]) > 10000
    # what the estimate buys on this chromosome: the heavier rank's share of the (exactly counted) window work
    zcut = _lib.lib().bmx_alpha_cut()
    w = distributed.block_work(gen, case.As, zcut, tg, 16)
    exact = np.zeros(M)
    for A in case.As:
        r = zcut / float(A)
        exact += np.searchsorted(gen, tg + r, 'right') - np.searchsorted(gen, tg - r, 'left')
    share = lambda parts: max(exact[p].sum() for p in parts) / exact.sum()
    rr, bal = share(distributed.assign(M, 2, 16)), share(distributed.assign(M, 2, 16, w))
    print('heavier rank holds %.1f %% of the work round-robin, %.1f %% balanced' % (100 * rr, 100 * bal))
    assert bal <= rr + 1e-12 and bal < 0.52
    with pytest.raises(ValueError):
        distributed.assign(M, 2, 16, w[:-1])
