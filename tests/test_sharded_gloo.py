"""N>1 path on CPU: world_size-2 (and 3) `gloo` process groups drive linear_programming_solver_amd.sharded
with a numpy shard engine; the result must equal the single-process fp64 oracle bit for bit.  Also checks
the row partition and that bench.py's input generator does not depend on the sharding."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _lp(m, n, seed):
    rng = np.random.default_rng(seed)
    return rng.random((m, n)), (n / 4.0) * (1.0 + rng.random(m)), rng.random(n)


def _worker(rank, world, port, m, n, seed, budget, outdir, lookahead=False, block=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from linear_programming_solver_amd.sharded import DistExchange, row_block, sharded_simplex_loop
        from tests.np_shard_engine import NumpyShardEngine
        A, b, c = _lp(m, n, seed)
        r0, r1 = row_block(m, world, rank)
        eng = NumpyShardEngine(A[r0:r1], b[r0:r1], c, r0, m, world)
        status, pivots, _ = sharded_simplex_loop([eng], DistExchange(), max_pivots=budget, poll_every=5,
                                                 lookahead=lookahead, block=block)
        Al, bl, cl, v, perm = eng.read()
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), A=Al, b=bl, c=cl, v=v, perm=perm, status=status,
                 pivots=pivots, r0=r0, r1=r1)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,budget,lookahead,block", [
    (2, (37, 50), -1, False, 1), (2, (64, 40), 7, False, 1), (3, (50, 33), -1, False, 1), (2, (41, 30), -1, True, 1),
    (2, (64, 40), 7, True, 1), (2, (41, 30), -1, False, 4), (2, (64, 40), 7, False, 3)])
def test_gloo_sharded_loop_matches_oracle(tmp_path, oracle, world, shape, budget, lookahead, block):
    m, n = shape
    seed = 100 + m
    port = _free_port()
    mp.spawn(_worker, args=(world, port, m, n, seed, budget, str(tmp_path), lookahead, block), nprocs=world, join=True)
    A, b, c = _lp(m, n, seed)
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    want = ref.simplex_loop(max_pivots=budget)
    wA, wb, wc, wv, wperm = ref.read()
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        assert (int(z["status"]), int(z["pivots"])) == (want["status"], want["pivots"])
        r0, r1 = int(z["r0"]), int(z["r1"])
        assert np.array_equal(z["A"].view(np.uint64), wA[r0:r1].view(np.uint64))
        assert np.array_equal(z["b"].view(np.uint64), wb[r0:r1].view(np.uint64))
        assert np.array_equal(z["c"].view(np.uint64), wc.view(np.uint64))        # replicated
        assert float(z["v"]) == wv and list(z["perm"]) == list(wperm)            # replicated


def test_row_blocks_partition_like_the_reference():
    from linear_programming_solver_amd.sharded import row_block
    for m in (1, 7, 8, 1000, 32768):
        for g in (1, 2, 3, 4, 8):
            blocks = [row_block(m, g, r) for r in range(g)]
            assert blocks[0][0] == 0 and blocks[-1][1] == m
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(g - 1))
            # LPState.java:222-223: from = k*m/THREADS, to = (k+1)*m/THREADS
            assert blocks == [((k * m) // g, ((k + 1) * m) // g) for k in range(g)]


def test_bench_generator_is_sharding_invariant():
    import bench
    m, n = 2500, 37
    A, b, c = bench.gen_rows(m, n, 3, 0, m)
    for g in (2, 3, 8):
        for r in range(g):
            r0, r1 = (r * m) // g, ((r + 1) * m) // g
            Al, bl, cl = bench.gen_rows(m, n, 3, r0, r1)
            assert np.array_equal(Al, A[r0:r1]) and np.array_equal(bl, b[r0:r1]) and np.array_equal(cl, c)


def test_local_exchange_two_shards_in_one_process(oracle):
    from linear_programming_solver_amd.sharded import LocalExchange, row_block, sharded_simplex_loop
    from tests.np_shard_engine import NumpyShardEngine
    m, n = 45, 61
    A, b, c = _lp(m, n, 9)
    engines = []
    for r in range(4):
        r0, r1 = row_block(m, 4, r)
        engines.append(NumpyShardEngine(A[r0:r1], b[r0:r1], c, r0, m, 4))
    status, pivots, _ = sharded_simplex_loop(engines, LocalExchange(), poll_every=3, lookahead=True)
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    want = ref.simplex_loop()
    assert (status, pivots) == (want["status"], want["pivots"])
    wA = ref.read()[0]
    got = np.vstack([e.read()[0] for e in engines])
    assert np.array_equal(got.view(np.uint64), wA.view(np.uint64))
