"""A DISTRIBUTION of reference trajectories for the Pano PSNR gate.

The 64-ray Pano training run of make_psnr_trace_pano.py is chaotic (ReLU-gate flips through the second-order path): one
fp32 and one fp64 trajectory do not say how far a change of summation order may move the held-out PSNR.  This script
trains the IMPORTED reference (fp32, CPU) K more times from initial weights that differ from the fixture's ONLY by one ulp
per element (random direction, PCG64(1000 + k)) - same batches, same three noise draws, same schedule - and stores the
held-out volume / surface PSNR and the loss trace of every run.  tests/test_gpu_psnr.py gates each kernel mode at
"within 0.1 dB of the reference's own min..max" over these runs plus the two of psnr_trace_pano.npz.
Build container only (needs /root/reference); arrays only are stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_psnr_ensemble_pano.py [K]
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def one(k):
    import torch
    import make_psnr_trace_pano as tp
    torch.set_num_threads(2)
    losses, psnr, psnr_s, _ = tp.train(torch.float32, perturb_seed=1000 + k)
    return k, losses, psnr, psnr_s


def main():
    import multiprocessing as mp
    import numpy as np
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    with mp.get_context("spawn").Pool(4) as pool:
        res = sorted(pool.map(one, range(K)))
    np.savez_compressed(os.path.join(HERE, "psnr_ensemble_pano.npz"),
                        seeds=np.array([1000 + k for k, *_ in res], np.int64),
                        losses=np.stack([r[1] for r in res]),
                        psnr=np.array([r[2] for r in res], np.float64),
                        psnr_surface=np.array([r[3] for r in res], np.float64))
    for k, l, p, s in res:
        print("run", k, "final loss", l[-1], "psnr", p, "surface", s)


if __name__ == "__main__":
    main()
