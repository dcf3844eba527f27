"""Developer check of the RCCL hook on a 1-GPU box: a torch.distributed 'nccl' group of ONE rank (all-reduce = identity)
drives a context that believes it is rank 0 of 2, so both all-reduces of the library go through RCCL on the device
buffers; the result must equal the plain single-rank evaluation of the same rows bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import gpz_amd, bench
from gpz_amd import dist as gdist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = dict(bench.CONFIGS["c4"]); cfg["n"] = 60000
model, theta, X, y, omega = bench.synth(cfg)
Xs, ys, oms, trs, _ = gdist.shard_rows(0, 2, X, y, omega)
a = gpz_amd.GPzContext(model, Xs, ys, None, oms, trs, None, rank=0, world=2, allreduce=gdist.make_allreduce())
fa, ga = a.eval(theta); fa2, ga2 = a.eval(theta); a.close()
b = gpz_amd.GPzContext(model, Xs, ys); fb, gb = b.eval(theta); b.close()
print("rccl-hook eval f=%.15g plain f=%.15g equal=%s grad equal=%s repeat equal=%s" % (fa, fb, fa == fb, np.array_equal(ga, gb), fa == fa2 and np.array_equal(ga, ga2)))
dist.destroy_process_group()
