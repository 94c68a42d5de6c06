#!/usr/bin/env python3
"""End-to-end fit on synthetic inputs, following the reference's main() (mft6.py:3450-3708) minus plotting:

    text model grid --spec_interpolator--> staged + broadened grid            (mft6.py:3512)
    data spectrum (made like mft6.py:3632-3642) + contrast / photometry inputs
    optimize_fit: nwalk random starts, lock-step chi^2 descent on the GPU       (mft6.py:3657)
    best third of the optimiser results seeds the ensemble                      (mft6.py:3668-3679)
    emcee-protocol sampling, walker state resident on the GPU                   (mft6.py:1490-1529)
    samples.txt

    python examples/fit_synthetic.py --out /tmp/fit --nwalk 48 --nstep 40 --nburn 50 --nsteps 300
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='fit_synthetic_out')
    ap.add_argument('--nwalk', type=int, default=48, help='optimiser start points (the best third become walkers)')
    ap.add_argument('--nstep', type=int, default=40, help='optimiser steps (param `nstep`)')
    ap.add_argument('--nburn', type=int, default=50)
    ap.add_argument('--nsteps', type=int, default=300)
    ap.add_argument('--seed', type=int, default=1)
    args = ap.parse_args()

    import mcmc_spec_amd.mft6 as gpu
    from mcmc_spec_amd import bands, loader, optimizer, staging, synth
    from mcmc_spec_amd.sampler import DeviceEnsembleSampler, run_reference_protocol

    os.makedirs(args.out, exist_ok=True)
    rng = np.random.default_rng(args.seed)
    t0 = time.time()

    # ---- "disk": a small BT-Settl-format text grid, isochrone, filters (all synthetic stand-ins) ----
    grid_dir = synth.write_btsettl_text_grid(os.path.join(args.out, 'BT-Settl_M-0.0a+0.0'),
                                             teffs=tuple(range(3000, 4300, 100)), loggs=(4.5, 5.0, 5.5), lo=5300.0,
                                             hi=9700.0, seed=5)
    matrix = synth.make_isochrone_matrix()
    spmin, spmax, res = 0.56, 0.89, 1700
    specs = loader.spec_interpolator([spmin * 1e4, spmax * 1e4], [3000, 4200], [4, 5.5], [5400, 9600], resolution=res,
                                     grid_dir=grid_dir, cache=os.path.join(args.out, 'grid_cache.npz'))
    print('grid: {} nodes x {} samples staged + broadened in {:.2f} s'.format(len(specs) - 1, len(specs['wl']),
                                                                              time.time() - t0))
    ctm = [[list(np.linspace(6000.0, 9500.0, 60))], [list(0.9 * np.ones(60))], [0], [7750.0]]
    ptm = [[], [], [], []]
    tmi, tma = 6000.0, 9500.0
    gpu.set_band_library(bands.make_bands(synth.synthetic_band_tables(), *synth.synthetic_vega()))
    gpu.set_av_prior(*synth.make_av_table())

    # ---- data: composite at the truth, resampled, 1 % noise (mft6.py:3632-3642), normalised (:3506-3507) ----
    truth = np.array([3850.0, 3325.0, 0.15, 0.52, 0.62, 2.0732e-3])
    wl_um = np.linspace(spmin + 0.002, spmax - 0.002, 2048)
    lg = staging.isochrone_logg(truth[:2], matrix)
    w1, c1, con, _, _ = gpu.make_composite(truth[:2], lg, truth[3:5], truth[5], ['x'], [], [wl_um.min(), wl_um.max()],
                                           specs, ctm, ptm, tmi, tma, None, nspec=2)
    c1 = c1 * 10.0 ** (-0.4 * truth[2] * specs.engine.ctx.ccm89_k(w1, 3.1))
    f = np.interp(wl_um * 1e4, w1, c1)
    d = f + rng.normal(0, 0.01 * f)
    data, err = [wl_um, d / np.median(d)], 0.01 * f / np.median(d)
    fr = [[con[0] + 0.01], [0.05], ['x'], [], [], []]
    plx, plx_err = 2.0732e-3, 0.0277e-3

    # ---- optimiser (mft6.py:3657) ----
    t1 = time.time()
    optimizer.optimize_fit(args.out, data, err, specs, args.nwalk, fr, [plx, plx_err], [0.106, 0.01], res, ctm, ptm, tmi,
                           tma, None, matrix, 288.456, 45.802, nspec=2, nstep=args.nstep, dist_fit=True, rad_prior=False,
                           seed=args.seed)
    chisqs, pars = np.genfromtxt(os.path.join(args.out, 'optimize_cs.txt')), np.genfromtxt(
        os.path.join(args.out, 'optimize_res.txt'))
    best = np.argsort(chisqs)[: int(len(chisqs) / 3)]  # mft6.py:3670-3674
    p0 = pars[best]
    if len(p0) % 2:
        p0 = p0[:-1]
    print('optimiser: {} starts x {} steps in {:.2f} s; best chi^2 {:.1f} at {}'.format(
        args.nwalk, args.nstep, time.time() - t1, chisqs.min(), np.round(pars[np.argmin(chisqs)], 4)))

    # ---- sampler (mft6.py:1490-1529), walker state on the device ----
    t2 = time.time()
    prior = [*np.zeros(10), plx, plx_err]  # mft6.py:3689
    eng = gpu._staged(specs, fr, 2, data, err, [wl_um.min(), wl_um.max()], ctm, ptm, tmi, tma, matrix, 3000.0, 4200.0, prior,
                      True, True, False, need_prior=True)
    sampler = DeviceEnsembleSampler(len(p0), 6, eng, seed=args.seed, chunk=50)
    samples = run_reference_protocol(sampler, p0, args.nburn, args.nsteps, nthin=50, dirname=args.out, fname='synthetic')
    print('sampler: {} walkers, {} + {} iterations in {:.2f} s; acceptance {:.2f}'.format(
        len(p0), args.nburn, sampler.iteration, time.time() - t2, sampler.acceptance_fraction.mean()))
    med = np.median(samples, axis=0)
    print('truth   :', truth)
    print('median  :', np.round(med, 5))
    print('wrote', os.path.join(args.out, 'samples.txt'), samples.shape)
    return truth, med, samples


if __name__ == '__main__':
    main()
