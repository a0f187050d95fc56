"""Repeat the one-rank RCCL "ingraph" capture test N times in fresh processes and print, per run, the verdict and the per-replay
(relative loss difference, relative gradient difference) between the captured data-parallel trainer and the plain captured trainer.
    python tools/ingraph_repeat.py [runs=12]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch.multiprocessing as mp  # noqa: E402

import test_gpu_dp as T  # noqa: E402

if __name__ == "__main__":
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    fails = 0
    for r in range(runs):
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(T._worker_ingraph, args=(1, T._free_port(), ret), nprocs=1, join=True)
        ok = ret.get(0) == "ok"
        fails += 0 if ok else 1
        print("run %2d: %s  diag (loss, grad) per replay: %s" % (r, "ok" if ok else ret.get(0), ret.get("diag")), flush=True)
    print("ingraph_repeat: %d / %d runs failed" % (fails, runs), flush=True)
    sys.exit(1 if fails else 0)
