"""World-size-2 gloo test of the N>1 path: chains are sharded across ranks with no
data-path collective; only summary totals are reduced.  The per-rank counting is done
by the oracle here (no GPU in this container); the sharding/reduction code is the
product's (plastid_amd/multigpu.py), the same bench.py uses with the nccl backend."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from plastid_amd import multigpu, synth
    from plastid_amd.packing import concat_file_major
    r, _, w = multigpu.init("gloo")
    assert (r, w) == (rank, world)
    genome, tx, reads, mapping = synth.make_config("C2", scale=0.0003, tx_scale=0.003)
    aln = concat_file_major([reads])
    mine = multigpu.shard_chains(tx.n, rank, world)
    sub = tx.subset(mine)
    p = sub.plan_arrays(rows=1)
    totals = []
    for kind, param in (("fiveprime", 12), ("center", 0)):
        spec = oracle.mapping_spec(kind, param)
        arrays, _ = oracle.count_segments(aln, spec, p["tid"], p["start"], p["end"], p["strand"])
        totals.append(sum(float(a.sum()) for a in arrays))
    n_int = multigpu.allreduce_int_totals([int(totals[0]), sub.n_positions, len(mine)])
    f_tot = multigpu.reduce_float_totals_ordered([totals[1]])
    t_max = multigpu.max_over_ranks(1.0 + rank)
    multigpu.barrier()
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.array(n_int + f_tot + [t_max, totals[1]], dtype=np.float64))


def test_two_rank_sharding_and_total_reduction(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "rank%d.npy" % r)) for r in range(world)]
    # every rank sees the same reduced totals
    assert np.array_equal(res[0][:5], res[1][:5])
    # and they equal the unsharded job
    from oracle import oracle
    from plastid_amd import multigpu, synth
    from plastid_amd.packing import concat_file_major
    genome, tx, reads, mapping = synth.make_config("C2", scale=0.0003, tx_scale=0.003)
    aln = concat_file_major([reads])
    p = tx.plan_arrays(rows=1)
    arrays, _ = oracle.count_segments(aln, oracle.mapping_spec("fiveprime", 12), p["tid"], p["start"], p["end"], p["strand"])
    assert int(res[0][0]) == int(sum(a.sum() for a in arrays))
    assert int(res[0][1]) == tx.n_positions and int(res[0][2]) == tx.n
    assert res[0][4] == 2.0  # max over ranks
    # float totals: fixed rank-order sum of the per-rank partials
    assert res[0][3] == res[0][5] + res[1][5]
    # shards are disjoint and cover all chains
    ids = np.concatenate([multigpu.shard_chains(tx.n, r, world) for r in range(world)])
    assert np.array_equal(ids, np.arange(tx.n))
