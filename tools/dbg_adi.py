import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench
from optconpy_amd import problems as pb, backend
import sadptprj_riclyap_adi.proj_ric_utils as pru
pr, tb, trct, ms = bench.build_inputs(58, 0.05, 16)
F = (-pr.A - pr.Nc).tocsr()
d = dict(pb.default_nwtn_adi_dict(), ms=ms, sweep_width=16, verbose=True, nwtn_max_steps=1)
out = pru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=tb, wmat=trct, nwtn_adi_dict=d)
print(out["adi_steps"])
