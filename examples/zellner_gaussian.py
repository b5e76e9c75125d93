#!/usr/bin/env python3
"""Gaussian-location coresets with outliers -- the reference's examples/zellner_gaussian/main.py
experiment (data recipe :33-54, algorithms :97-108, KL metrics :153-167) on the MI355X path.

    python examples/zellner_gaussian.py BCORES 1          # alg in {BCORES, BPSVI, SVI, GIGAO, GIGAR, RAND, PRIOR}, trial seed

Differences from the reference script: `import beta_cores_amd as bc`, the projectors are the device
ones (K1 on the GPU) and `weighted_post` is `bc.gaussian_weighted_post` (K4).  BPSVI is built for m = 1..M one after
the other instead of in a multiprocessing pool (main.py:126-135): every build starts from the RNG state the pool's
forked children would inherit.  Results are printed, not pickled.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import beta_cores_amd as bc


def gaussian_KL(mu0, Sig0, mu1, Sig1inv):
    t1 = np.dot(Sig1inv, Sig0).trace()
    t2 = np.dot((mu1 - mu0), np.dot(Sig1inv, mu1 - mu0))
    t3 = -np.linalg.slogdet(Sig1inv)[1] - np.linalg.slogdet(Sig0)[1]
    return 0.5 * (t1 + t2 + t3 - mu0.shape[0])


def run(nm='BCORES', tr=1, N=5000, d=50, M=40, opt_itrs=200, n_subsample_opt=200, n_subsample_select=1000, proj_dim=200,
        pihat_noise=0.75, i0=0.1, verbose=True):
    """One trial of the experiment.  Statement order follows the reference script (main.py:15-108), so with the same
    seed the global NumPy RNG stream -- data, the four projectors' constructor draws, the noise of the 'realistic'
    tangent space -- is the script's.  Returns dict(w, p, idcs (per m = 0..M), rkl, fkl, Xc)."""
    np.random.seed(tr)
    mu0, Sig0 = np.zeros(d), np.eye(d)
    Sig = 500 * np.eye(d)
    th = np.zeros(d)
    Sig0inv, Siginv = np.linalg.inv(Sig0), np.linalg.inv(Sig)
    logdetSig = np.linalg.slogdet(Sig)[1]
    X = np.random.multivariate_normal(th, Sig, N)
    mup, LSigp, LSigpInv = bc.gaussian_weighted_post(mu0, Sig0inv, Siginv, X, np.ones(X.shape[0]))   # clean-data posterior
    Sigp, SigpInv = LSigp.dot(LSigp.T), LSigpInv.dot(LSigpInv.T)
    Xc = np.concatenate((X, np.random.multivariate_normal(th + 200, 0.5 * Sig, int(N / 50.)),
                         np.random.multivariate_normal(th + 150, 0.1 * Sig, int(N / 50.)),
                         np.random.multivariate_normal(th, 10 * Sig, int(N / 10.))))

    model = bc.likelihoods.GaussianLocation(Siginv, logdetSig)
    # every projector draws its first Theta when constructed (projector.py:18,46): all four are built, in the
    # script's order, whichever algorithm runs
    sampler_optimal = lambda n, w, pts: mup + np.random.randn(n, mup.shape[0]).dot(LSigp.T)
    prj_optimal = bc.DeviceProjector(sampler_optimal, proj_dim, model)
    U = np.random.rand()
    muhat = U * mup + (1. - U) * mu0
    Sighat = U * Sigp + (1. - U) * Sig0
    muhat += pihat_noise * np.sqrt((muhat ** 2).sum()) * np.random.randn(muhat.shape[0])
    Sighat *= np.exp(-2 * pihat_noise * np.fabs(np.random.randn()))
    LSighat = np.linalg.cholesky(Sighat)
    sampler_realistic = lambda n, w, pts: mup + np.random.randn(n, mup.shape[0]).dot(LSighat.T)
    prj_realistic = bc.DeviceProjector(sampler_realistic, proj_dim, model)

    def sampler_w(sz, wts, pts):
        if pts.shape[0] == 0:
            wts, pts = np.zeros(1), np.zeros((1, Xc.shape[1]))
        muw, LSigw, _ = bc.gaussian_weighted_post(mu0, Sig0inv, Siginv, pts, wts)
        return muw + np.random.randn(sz, muw.shape[0]).dot(LSigw.T)

    prj_w = bc.DeviceProjector(sampler_w, proj_dim, model)
    prj_bw = bc.DeviceBetaProjector(sampler_w, proj_dim, model)

    sched = lambda i: i0 / (1. + i)
    if nm == 'BCORES':
        alg = bc.BetaCoreset(Xc, prj_bw, opt_itrs=opt_itrs, n_subsample_opt=n_subsample_opt,
                             n_subsample_select=n_subsample_select, step_sched=sched, beta=.1, learn_beta=False)
    elif nm == 'SVI':
        alg = bc.SparseVICoreset(Xc, prj_w, opt_itrs=opt_itrs, n_subsample_opt=n_subsample_opt,
                                 n_subsample_select=n_subsample_select, step_sched=sched)
    elif nm == 'GIGAO':
        alg = bc.HilbertCoreset(Xc, prj_optimal)
    elif nm == 'GIGAR':
        alg = bc.HilbertCoreset(Xc, prj_realistic)
    elif nm == 'BPSVI':
        alg = bc.BatchPSVICoreset(Xc, prj_w, opt_itrs=opt_itrs, n_subsample_opt=n_subsample_opt,
                                  step_sched=lambda m: lambda i: i0 / (1. + i))
    elif nm == 'RAND':
        alg = bc.UniformSamplingCoreset(Xc)
    elif nm == 'PRIOR':
        alg = None
    else:
        raise SystemExit('alg must be one of BCORES, BPSVI, SVI, GIGAO, GIGAR, RAND, PRIOR')

    w, p, idl = [np.array([0.])], [np.zeros((1, Xc.shape[1]))], [np.zeros(0, dtype=np.int64)]
    fork_state = np.random.get_state()
    for m in range(1, M + 1):
        if nm == 'PRIOR':
            w.append(np.array([0.]))
            p.append(np.zeros((1, Xc.shape[1])))
            idl.append(np.zeros(0, dtype=np.int64))
            continue
        if nm == 'BPSVI':
            np.random.set_state(fork_state)          # main.py:126-135: each m is built in a forked child of the pool
        alg.build(1, m)
        got = alg.get()
        w.append(got[0].copy())
        p.append(got[1].copy())
        idl.append(got[2].copy())
    if nm == 'BPSVI':
        np.random.set_state(fork_state)              # the parent's own stream never moved
    rkl, fkl = np.zeros(M + 1), np.zeros(M + 1)
    if verbose:
        print('%4s %12s %12s' % ('m', 'reverse KL', 'forward KL'))
    for m in range(M + 1):
        muw, LSigw, LSigwInv = bc.gaussian_weighted_post(mu0, Sig0inv, Siginv, p[m], w[m])
        Sigw = LSigw.dot(LSigw.T)
        rkl[m] = gaussian_KL(muw, Sigw, mup, SigpInv)
        fkl[m] = gaussian_KL(mup, Sigp, muw, LSigwInv.dot(LSigwInv.T))
        if verbose and (m in (1, 2, 5, 10, 20, M) or m % 50 == 0):
            print('%4d %12.4f %12.4f' % (m, rkl[m], fkl[m]))
    return dict(w=w, p=p, idcs=idl, rkl=rkl, fkl=fkl, Xc=Xc)


def main():
    nm = sys.argv[1] if len(sys.argv) > 1 else 'BCORES'
    tr = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    run(nm, tr, N=int(os.environ.get('N', 5000)), d=int(os.environ.get('D', 50)), M=int(os.environ.get('M', 40)))


if __name__ == '__main__':
    main()
