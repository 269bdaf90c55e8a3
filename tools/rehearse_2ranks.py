"""Rehearsal of the multi-rank shift-parallel path on ONE GPU: 2 processes on device 0,
gloo backend (NCCL refuses two ranks on one device), device tensors through HipOps.
    python tools/rehearse_2ranks.py"""
import os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, torch.distributed as dist, torch.multiprocessing as mp


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from optconpy_amd import _lib, problems as pb
    from optconpy_amd.shift_parallel import HipOps, lyap_adi_shift_parallel
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    from make_golden import cfg1_inputs
    pr, tb, trct, ms = cfg1_inputs()
    F = (-pr.A - pr.Nc).tocsr()
    ctxs = []
    for _ in range(2):
        c = _lib.Context(0)
        c.set_operator(F.T.tocsr(), pr.M.T.tocsr(), pr.J)
        ctxs.append(c)
    ops = HipOps(ctxs[0], ctxs[1:])
    W = ops.to_panel(trct)
    blocks, info = lyap_adi_shift_parallel(ops, ms, W, adi_max_steps=200, adi_newZ_reltol=1e-8, width=4)
    Z = torch.cat(blocks, dim=1).contiguous()
    K = ops.gain(-1.0, Z, ops.to_panel(tb.toarray())).cpu().numpy()
    np.save(os.path.join(out, "K%d.npy" % rank), K)
    np.save(os.path.join(out, "info%d.npy" % rank),
            np.array([info["adi_steps"], info["sweeps"], ops.shift_solves]))
    dist.destroy_process_group()


if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = os.environ.get("TMPDIR", "/tmp")
    mp.spawn(worker, args=(2, port, out), nprocs=2, join=True)
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cfg1_golden.npz"))
    for r in range(2):
        K = np.load(os.path.join(out, "K%d.npy" % r))
        steps, sweeps, solves = np.load(os.path.join(out, "info%d.npy" % r))
        print("rank %d: %d ADI steps in %d sweeps, local solves %d" % (r, steps, sweeps, solves))
        print("rank %d K rel diff vs golden %.2e" % (r, np.linalg.norm(K - g["K_lyap"]) / np.linalg.norm(g["K_lyap"])))
