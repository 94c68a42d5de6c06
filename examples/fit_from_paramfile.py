#!/usr/bin/env python3
"""A fit driven by a parameter file in the reference's format (``param_koi2298.txt``), following ``main`` of the
reference (mft6.py:3450-3708) minus plotting -- BASELINE config 1 / 5 plumbing:

    parameter file --params.load_run--> data (read, telluric mask, crop, normalise), fr, prior   (mft6.py:3458-3575)
    model grid (here: a synthetic BT-Settl-format text grid) --spec_interpolator--> staged grid   (mft6.py:3512)
    emcee-protocol sampling of logposterior with the reference's args / kwargs                    (mft6.py:1490-1529)

    python examples/fit_from_paramfile.py -f my_param.txt [--data-root DIR] [--out DIR] [--nwalkers 32] [--steps 100]

With no ``-f`` a small parameter file and a synthetic spectrum are written first (so the example runs anywhere).
For a batch of targets (BASELINE config 5: one target per GPU) launch one process per GPU, each with its own
parameter file:  ``python -m torch.distributed.run --nproc-per-node 8 ... fit_from_paramfile.py -f 'param_{rank}.txt'``
(``{rank}`` is replaced by $RANK; the processes never communicate: replicas, no collective).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

DEMO = """### demo parameter file (the reference's format: `key value`, one space, value ends at the first tab)
models btsettl\t\t#model set
dirname demo\t\t#output directory
fname demo\t\t#chain file name
res 1700\t\t#spectral resolution
tmin 3000\t\t#K
tmax 4200\t\t#K
specmin 5400\t\t#A
specmax 9600\t\t#A
mask False
rad_prior False
filename demo_spectrum.txt\t#three columns: wavelength [um], flux, error
spmin 0.57\t\t#um
spmax 0.88\t\t#um
cmag [2.4]\t\t#contrast magnitudes
cerr [0.05]\t\t#their errors
cfilt ['x']\t\t#contrast filters
pmag []\t\t#unresolved magnitudes
perr []\t\t#
pfilt []\t\t#
plx 2.0732e-3\t\t#arcsec
plx_err 0.0277e-3\t\t#arcsec
dist_fit True
av 0.106\t\t#mag
av_err 0.01\t\t#mag
ra 288.456118
dec 45.802226
nwalk 96\t\t#optimiser starts
nstep 40\t\t#optimiser steps
nspec 2\t\t#stars
ndust 0\t\t#disks
nburn 50\t\t#burn-in
nsteps 200\t\t#steps
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-f', '--file', default=None, help="parameter file ('{rank}' is replaced by $RANK)")
    ap.add_argument('--data-root', default=None)
    ap.add_argument('--out', default='fit_paramfile_out')
    ap.add_argument('--nwalkers', type=int, default=32)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--seed', type=int, default=1)
    args = ap.parse_args()

    import mcmc_spec_amd.mft6 as gpu
    from mcmc_spec_amd import bands, loader, params, staging, synth
    from mcmc_spec_amd.sampler import EnsembleSampler

    rank = int(os.environ.get('RANK', '0'))
    gpu.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    out = args.out if 'RANK' not in os.environ else '{}_{}'.format(args.out, rank)
    os.makedirs(out, exist_ok=True)
    matrix = synth.make_isochrone_matrix()
    grid_dir = synth.write_btsettl_text_grid(os.path.join(out, 'BT-Settl_M-0.0a+0.0'), teffs=tuple(range(3000, 4300, 100)),
                                             loggs=(4.5, 5.0, 5.5), lo=5300.0, hi=9700.0, seed=5)
    ctm = [[list(np.linspace(6000.0, 9500.0, 60))], [list(0.9 * np.ones(60))], [0], [7750.0]]   # one flat contrast filter 'x'
    ptm = [[], [], [], []]
    tmi, tma = 6000.0, 9500.0
    gpu.set_band_library(bands.make_bands(synth.synthetic_band_tables(), *synth.synthetic_vega()))
    gpu.set_av_prior(*synth.make_av_table())

    if args.file is None:
        # a synthetic target: the composite at a known truth, 1 % noise, written as the three-column file the
        # reference reads (mft6.py:3492), in physical flux units so that the median normalisation has work to do
        specs0 = loader.spec_interpolator([5700.0, 8800.0], [3000, 4200], [4, 5.5], [5400, 9600], resolution=1700,
                                          grid_dir=grid_dir)
        truth = np.array([3850.0, 3325.0, 0.15, 0.52, 0.62, 2.0732e-3])
        wl_um = np.linspace(0.565, 0.885, 1500)
        lg = staging.isochrone_logg(truth[:2], matrix)
        w1, c1, con, _, _ = gpu.make_composite(truth[:2], lg, truth[3:5], truth[5], ['x'], [], [wl_um.min(), wl_um.max()],
                                               specs0, ctm, ptm, tmi, tma, None, nspec=2)
        c1 = c1 * 10.0 ** (-0.4 * truth[2] * specs0.engine.ctx.ccm89_k(w1, 3.1))
        f = np.interp(wl_um * 1e4, w1, c1)
        rng = np.random.default_rng(args.seed)
        np.savetxt(os.path.join(out, 'demo_spectrum.txt'), np.column_stack((wl_um, f + rng.normal(0, 0.01 * f), 0.01 * f)))
        parfile = os.path.join(out, 'param_demo.txt')
        open(parfile, 'w').write(DEMO.replace('cmag [2.4]', 'cmag [{:.4f}]'.format(con[0])))
        data_root = out
        print('wrote', parfile, '(truth: {})'.format(truth))
    else:
        parfile, data_root = args.file.replace('{rank}', str(rank)), args.data_root

    run = params.load_run(parfile, data_root=data_root)                        # mft6.py:3458-3575
    print('{}: {} px after the ({}, {}) um crop{}, nspec {}, dist_fit {}, rad_prior {}'.format(
        os.path.basename(run.filename), len(run.err), run.spmin, run.spmax, ' + telluric mask' if run.mask else '',
        run.nspec, run.dist_fit, run.rad_prior))
    specs = loader.spec_interpolator([run.spmin * 1e4, run.spmax * 1e4], [run.tmin, run.tmax], run.logg_range,
                                     run.specrange, resolution=run.res, grid_dir=grid_dir)   # mft6.py:3512-3513
    ndim = 2 * run.nspec + 2
    # the reference's emcee call, argument for argument (mft6.py:1491-1492, 3688-3689)
    sargs = [run.fr, run.nspec, run.ndust, run.data, run.err, run.res, run.r, specs, ctm, ptm, tmi, tma, None,
             float(run.tmin), float(run.tmax), matrix, run.ra, run.dec]
    skw = {'dust': False, 'norm': True, 'prior': run.prior(ndim), 'a': True, 'models': run.models,
           'dist_fit': run.dist_fit, 'rad_prior': run.rad_prior}
    sampler = EnsembleSampler(args.nwalkers, ndim, gpu.logposterior, args=sargs, kwargs=skw, vectorize=True, seed=args.seed)
    rng = np.random.default_rng(args.seed + 1)
    centre = np.array([3800.0, 3300.0, run.av, 0.5, 0.6, run.plx])
    p0 = centre + rng.normal(size=(args.nwalkers, ndim)) * np.array([40.0, 40.0, 0.01, 0.02, 0.02, run.plx_err])
    st = sampler.run_mcmc(p0, args.steps)
    flat = sampler.get_chain(flat=True, discard=args.steps // 2)
    print('{} walkers x {} steps; acceptance {:.2f}; log posterior of the last ensemble {:.1f} .. {:.1f}'.format(
        args.nwalkers, args.steps, sampler.acceptance_fraction.mean(), st.log_prob.min(), st.log_prob.max()))
    print('posterior medians:', np.round(np.median(flat, axis=0), 5))
    np.savetxt(os.path.join(out, 'samples.txt'), sampler.chain.reshape(-1, ndim))
    return sampler


if __name__ == '__main__':
    main()
