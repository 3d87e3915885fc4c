import os, sys, socket, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from functools import partial

CASE = dict(M=768, N=900, K=24, unity=True, its=9)

def worker(rank, world, port, mode, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch, torch.distributed as dist
    import proxmin_amd as pm
    from proxmin_amd import distributed as pdist
    from oracle import nmf_oracle as orc
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pm.set_default_mode(mode)
    c = CASE
    Y, A0, S0 = orc.synthetic_problem(c["M"], c["N"], c["K"], np.float32, unity_S=True, seed=4)
    r0, r1 = pdist.shard_rows(c["M"], world)[rank]
    A_l, S = A0[r0:r1].copy(), S0.copy()
    conv, n = pdist.nmf_adaprox_sharded(Y[r0:r1], A_l, S, c["M"], prox_A=pm.operators.prox_plus, prox_S=partial(pm.operators.prox_unity_plus, axis=0),
                                        scheme="amsgrad", check_convergence=False, e_rel=1e-3, max_iter=c["its"])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), A=A_l, S=S, n=n, r0=r0, r1=r1)
    dist.destroy_process_group()

def main():
    import torch.multiprocessing as mp
    import proxmin_amd as pm
    from oracle import nmf_oracle as orc
    mode = sys.argv[1]
    c = CASE
    Y, A0, S0 = orc.synthetic_problem(c["M"], c["N"], c["K"], np.float32, unity_S=True, seed=4)
    pm.set_default_mode(mode)
    A1, S1 = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, A1, S1, algorithm=pm.adaprox, scheme="amsgrad", prox_S=partial(pm.operators.prox_unity_plus, axis=0), max_iter=c["its"], e_rel=1e-3, check_convergence=False)
    for rep in range(int(sys.argv[2])):
        d = tempfile.mkdtemp()
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context("spawn")
        procs = [ctx.Process(target=worker, args=(r, 2, port, mode, d)) for r in range(2)]
        [p.start() for p in procs]; [p.join(60) for p in procs]
        if any(p.exitcode != 0 for p in procs):
            print(rep, "exit codes", [p.exitcode for p in procs], flush=True)
            [p.kill() for p in procs if p.exitcode is None]
            continue
        z = [np.load(os.path.join(d, "rank%d.npz" % r)) for r in range(2)]
        print(rep, "n", [int(x["n"]) for x in z], "S err %.2e %.2e" % tuple(np.abs(x["S"] - S1).max() for x in z),
              "ranks equal", np.array_equal(z[0]["S"], z[1]["S"]),
              "A err %.2e %.2e" % tuple(np.abs(x["A"] - A1[int(x["r0"]):int(x["r1"])]).max() for x in z), flush=True)

if __name__ == "__main__":
    main()
