"""The RCCL branch of the multi-GPU path on real hardware, in a world of ONE: `init_process_group("nccl", world_size=1)`,
then the zero-copy torch view of the engine's device buffer (`multigpu.device_tensor` over `pc_counts_device_ptr`) goes
through `dist.all_reduce` -- what `bench.py --gpus N` does with the per-chain sums (bench.py, run_partitioned), minus
the other ranks.  Run in a child process: the process group is the child's."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r"""
import os, socket, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
from plastid_amd import multigpu, synth
from plastid_amd.engine import Engine

with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%%d" %% port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
genome, tx, reads, _ = synth.make_config("C4", scale=0.0005, tx_scale=0.01)
eng = Engine(0)
eng.set_alignments([reads])
synth.mapping_factory(("fiveprime", 12))._configure(eng)
p = tx.plan_arrays(rows=1)
gp = multigpu.GenomePartition([reads], p, 1)
seg_chain = np.repeat(np.arange(tx.n, dtype=np.int64), np.diff(tx.ex_off))
sp = gp.chain_sum_plan_arrays(0, seg_chain, tx.n, 1)
plan = eng.plan(sp["tid"], sp["start"], sp["end"], sp["strand"], sp["out_off"], sp["out_step"], sp["row_stride"], sp["out_elems"], 1)
plan.launch(np.int64)
eng.sync()
want = plan.read().copy()
assert want.sum() > 0
torch.cuda.synchronize()
t = multigpu.allreduce_device_sums(plan.device_ptr, tx.n, "int64", force=True)      # RCCL on the engine's own buffer
torch.cuda.synchronize()
got = t.cpu().numpy()
assert np.array_equal(got, want), "a sum over one rank changed the values"
assert np.array_equal(plan.read(), want), "the engine's buffer is what RCCL reduced in place"
# the small totals of the bench line travel the same way
assert multigpu.allreduce_int_totals([3, 4], device="cuda") == [3, 4]
assert multigpu.reduce_float_totals_ordered([0.25], device="cuda") == [0.25]
assert multigpu.max_over_ranks(1.5, device="cuda") == 1.5
multigpu.barrier()
plan.close(); eng.close()
dist.destroy_process_group()
print("rccl world-of-one ok: %%d chain sums" %% tx.n)
"""


def test_rccl_allreduce_on_the_engines_device_buffer(tmp_path):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    script = tmp_path / "child.py"
    script.write_text(CHILD % {"root": ROOT})
    proc = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert proc.returncode == 0, proc.stderr.decode()[-3000:]
    assert b"rccl world-of-one ok" in proc.stdout
