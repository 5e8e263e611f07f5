import os, sys, time, tempfile
ROOT = "/root/repo" if os.path.isdir("/root/repo/plastid_amd") else os.getcwd()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ROOT))
import torch
torch.cuda.set_device(0)
x = torch.zeros(16, device="cuda"); torch.cuda.synchronize()
from plastid_amd import synth
from plastid_amd.engine import Engine
from tests import bam_writer
n = int(float(sys.argv[1]))
genome, tx, reads, mapping = synth.make_config("C2", scale=n / 1e8)
tmp = tempfile.mkdtemp(prefix="pc_bamres_")
path = os.path.join(tmp, "s.bam")
bam_writer.write_bam_realistic(path, reads, threads=16)
def leg(tag):
    eng = Engine(0)
    for rep in range(3):
        sys.stderr.write("== %s pass %d\n" % (tag, rep)); sys.stderr.flush()
        eng.clear_alignments()
        t1 = time.perf_counter()
        eng.add_bam(path)
        eng.sync()
        t2 = time.perf_counter()
        print("%s pass %d: file -> staged %.1f ms" % (tag, rep, (t2 - t1) * 1e3), flush=True)
    eng.close()
leg("torch-cuda")
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
dist.all_reduce(x); torch.cuda.synchronize()
leg("torch-nccl")
os.environ["PC_BAM_STREAMS"] = "3"
leg("torch-nccl-3streams")
os.remove(path); os.rmdir(tmp)
