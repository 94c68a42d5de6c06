"""Host-side staging: turn the reference's Python arguments into the dense tables the HIP library
consumes (``struct msx_problem``, include/msx.h).

Nothing here touches model fluxes: the grid goes to HBM once (``stage_specs``) and every
flux-dependent table (pixel pair table, per-node band integrals, CCM89 curve, broadened nodes) is
produced on the device by ``libmsx.so``.  What *is* computed here are the walker-independent index
and weight tables, with the same NumPy/SciPy expressions the reference uses so that masks and
``searchsorted`` decisions are bit-identical (SURVEY.md §7 "bit-level staging semantics").
"""
from __future__ import annotations

import ctypes as C

import numpy as np
from numpy.polynomial import polyutils as pu
from scipy.interpolate import interp1d

from . import _lib
from .bands import BAND_NAMES_3, BAND_NAMES_6, _trapz_weights


def parse_specs(specs):
    """``specs`` dict (mft6.py:363,383) -> (teff_nodes, logg_nodes, wl, flux[nt][ng][nwl], present).

    Node values come from the dict keys (the reference re-lists the model directory instead,
    mft6.py:423-436).  Absent (T, g) combinations are marked not-present: touching
    one raises KeyError at evaluation time like the reference (mft6.py:489-500)."""
    ts, gs = set(), set()
    for k in specs:
        if k == 'wl':
            continue
        a, b = k.split(', ')
        ts.add(int(a))
        gs.add(float(b))
    teff = sorted(ts)
    logg = sorted(gs)
    wl = np.asarray(specs['wl'], dtype=np.float64)
    flux = np.zeros((len(teff), len(logg), len(wl)))
    present = np.zeros((len(teff), len(logg)), dtype=np.uint8)
    for it, t in enumerate(teff):
        for ig, g in enumerate(logg):
            v = specs.get('{}, {}'.format(t, g))
            if v is not None:
                flux[it, ig] = v
                present[it, ig] = 1
    return np.array(teff, dtype=float), np.array(logg, dtype=float), wl, flux, present


def composite_window_um(r, tmi, tma, ctm, ptm):
    """The window make_composite passes to get_spec, in micron (mft6.py:663-673,687)."""
    wlmin, wlmax = np.inf, 0
    for w in list(ctm[0]) + list(ptm[0]):
        if min(w) < wlmin:
            wlmin = min(w)
        if max(w) > wlmax:
            wlmax = max(w)
    return [min(min(r), tmi / 1e4, wlmin / 1e4) - 1e-4, max(max(r), tma / 1e4, wlmax / 1e4) + 1e-4]


def window_slice(wl, reg_um):
    """Contiguous [j0, j0+n) selected by get_spec's crop masks (mft6.py:537-542)."""
    lim = np.array(reg_um) * 1e4
    keep = np.where((wl >= min(lim)) & (wl <= max(lim)))[0]
    if keep.size == 0:
        raise ValueError('the composite window selects no model samples')
    if keep[-1] - keep[0] + 1 != keep.size:
        raise ValueError('model wavelength grid must be sorted')
    return int(keep[0]), int(keep.size)


def resample_tables(wave, x):
    """np.interp / interp1d(kind='linear') bracket of each query (mft6.py:1169-1170).

    Returns (lo, t) with result = y[lo] + (y[lo+1]-y[lo])*t.  Raises the same ValueError scipy does
    when a query lies outside [wave[0], wave[-1]]."""
    if np.any(np.diff(wave) <= 0):
        raise ValueError('model wavelength grid must be strictly increasing')
    if np.any(x < wave[0]):
        raise ValueError("A value ({}) in x_new is below the interpolation range's minimum value ({})."
                         .format(x.min(), wave[0]))
    if np.any(x > wave[-1]):
        raise ValueError("A value ({}) in x_new is above the interpolation range's maximum value ({})."
                         .format(x.max(), wave[-1]))
    j = np.searchsorted(wave, x, side='right') - 1
    last = j >= len(wave) - 1
    j = np.where(last, len(wave) - 2, j)
    t = (x - wave[j]) / (wave[j + 1] - wave[j])
    t = np.where(last, 1.0, t)
    return j.astype(np.int64), t


def contrast_weights(wave, ran, tm):
    """Weights for one contrast filter: ``trapz(s*tran, w)`` over the in-band samples (mft6.py:719-731)."""
    ran = np.asarray(ran, dtype=float)
    inband = np.where((wave <= max(ran)) & (wave >= min(ran)))[0]
    if inband.size < 2:
        raise ValueError('a contrast filter does not overlap the model window')
    w = wave[inband]
    tran = interp1d(ran, tm)(w)
    return int(inband[0]), _trapz_weights(w) * tran


def sorted_isochrone(matrix):
    """1 Gyr rows (``matrix[:,1] == 9.0``, first 220) sorted by Teff the way interp1d sorts its x
    (stable mergesort) (mft6.py:73-76, 90-95)."""
    sel = np.where(np.asarray(matrix[:, 1]) == 9.0)[0][:220]
    x = np.asarray(matrix[sel, 4], dtype=float)
    order = np.argsort(x, kind='mergesort')
    return x[order], np.asarray(matrix[sel, 5], dtype=float)[order], np.asarray(matrix[sel, 6], dtype=float)[order]


def prior_vectors(prior, nspec, dist_fit=True):
    """The reference's ``prior`` list re-ordered to parameter order: mft6.py:1243-1254 with ``dist_fit``
    (Teff.., A_V, radii.., parallax), :1303-1312 without (Teff.., A_V and ``nspec-1`` radius entries only --
    the slices there are one element shorter and the parallax has no Gaussian term)."""
    ndim = 2 * nspec + 2
    mean, sig = np.zeros(_lib.MAX_DIM), np.ones(_lib.MAX_DIM)
    if isinstance(prior, (int, float)) and prior == 0:
        return mean, sig, 0
    prior = [float(p) for p in prior]
    if dist_fit:
        ps = prior[:nspec] + [prior[2 * nspec]] + prior[2 * nspec + 2:3 * nspec + 2] + [prior[-2]]
        ss = prior[nspec:2 * nspec] + [prior[2 * nspec + 1]] + prior[3 * nspec + 2:4 * nspec + 2] + [prior[-1]]
    else:
        ps = prior[:nspec] + [prior[2 * nspec]] + prior[2 * nspec + 2:3 * nspec + 1]
        ss = prior[nspec:2 * nspec] + [prior[2 * nspec + 1]] + prior[3 * nspec + 2:4 * nspec + 1]
    assert len(ps) <= ndim
    mean[:len(ps)] = ps
    sig[:len(ss)] = ss
    return mean, sig, 1


class StagedTables:
    """Owns the NumPy arrays referenced by a ``MsxProblem`` (keeps them alive for the ctypes call)."""

    def __init__(self):
        self.keep = []
        self.prob = _lib.MsxProblem()

    def f64(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.size == 0:
            a = np.zeros(1)
        self.keep.append(a)
        return _lib.dptr(a)

    def i64(self, a):
        a = np.ascontiguousarray(a, dtype=np.int64)
        if a.size == 0:
            a = np.zeros(1, dtype=np.int64)
        self.keep.append(a)
        return _lib.iptr(a)


def build_problem(ctx, grid_wl, data, err, fr, r, ctm, ptm, tmi, tma, matrix, nspec=2, bands=None,
                  av_table=None, tmin=-np.inf, tmax=np.inf, prior=0, use_av=True, dist_fit=True, rad_prior=False,
                  spectrum=True):
    """Assemble ``struct msx_problem`` for one dataset.  Arguments follow ``logposterior``'s
    (mft6.py:1459): ``data = [wl_um, flux]``, ``fr = [cmag, cerr, cfilt, pmag, perr, pfilt]``,
    ``ctm``/``ptm`` = ``[wls, tras, n_res_el, cwl]``.  ``bands`` maps pyphot-style names to
    ``bands.Band``; ``av_table`` = (edges_pc, mu, sigma) or None."""
    st = StagedTables()
    P = st.prob
    wl_um, flux = np.asarray(data[0], dtype=float), np.asarray(data[1], dtype=float)
    err = np.asarray(err, dtype=float)
    if not (len(wl_um) == len(flux) == len(err)):
        raise ValueError('Data and model must be the same length')
    npix = len(wl_um)
    grid_wl = np.asarray(grid_wl, dtype=float)

    reg = composite_window_um(r, tmi, tma, ctm, ptm)
    j0, nwin = window_slice(grid_wl, reg)
    wave = grid_wl[j0:j0 + nwin]

    lo, t = resample_tables(wave, wl_um * 1e4)  # mft6.py:1170: intep(wl * 1e4)
    dom = pu.getdomain(wl_um)  # Polynomial.fit(wl, ...) maps its domain to [-1, 1]
    u = pu.mapdomain(wl_um, dom, np.array([-1.0, 1.0]))
    V = np.stack([np.ones_like(u), u, u * u], axis=1)
    minv = np.linalg.inv(V.T @ V)

    P.struct_size = C.sizeof(_lib.MsxProblem)
    P.nspec = int(nspec)
    P.npix = npix
    P.pix_lo = st.i64(lo + j0)
    P.pix_t = st.f64(t)
    P.pix_u = st.f64(u)
    P.pix_flux = st.f64(flux)
    P.pix_err = st.f64(err)
    P.median_flux = float(np.median(flux))
    for i in range(9):
        P.fit_minv[i] = float(minv.flat[i])

    cmag, cerr = [float(x) for x in fr[0]], [float(x) for x in fr[1]]
    nc = len(fr[2])
    pmag, perr = [float(x) for x in fr[3]], [float(x) for x in fr[4]]
    nph = len(fr[5])
    if nc > _lib.MAX_BANDS or nph > _lib.MAX_BANDS:
        raise ValueError('at most {} contrast filters and {} photometric bands'.format(_lib.MAX_BANDS, _lib.MAX_BANDS))
    i0s, lens, ws = [], [], []
    for n in range(nc):
        i0, w = contrast_weights(wave, ctm[0][n], ctm[1][n])
        i0s.append(i0 + j0)
        lens.append(len(w))
        ws.append(w)
    zero, phot_cwl = [], []
    if nph:
        if bands is None:
            raise ValueError('photometric bands requested but no band library given (see mcmc_spec_amd.bands)')
        names = BAND_NAMES_3 if nph == 3 else BAND_NAMES_6  # mft6.py:766-769
        for n in range(nph):
            b = bands[names[n]]
            i0, w = b.weights_on(wave)
            i0s.append(i0 + j0)
            lens.append(len(w))
            ws.append(w)
            zero.append(b.zero_flux)
        phot_cwl = [float(p) for p in ptm[3]][:nph]  # mft6.py:831
    P.n_contrast, P.n_phot = nc, nph
    P.band_i0 = st.i64(i0s)
    P.band_len = st.i64(lens)
    P.band_w = st.f64(np.concatenate(ws) if ws else np.zeros(0))
    P.cmag, P.cerr = st.f64(cmag), st.f64(cerr)
    P.pmag, P.perr = st.f64(pmag), st.f64(perr)
    P.phot_zero = st.f64(zero)
    P.phot_k = st.f64(ctx.ccm89_k(np.array(phot_cwl), 3.1) if nph else np.zeros(0))
    P.win_j0, P.win_n = j0, nwin

    it, ig, il = sorted_isochrone(np.asarray(matrix))
    P.niso = len(it)
    P.iso_teff, P.iso_logg, P.iso_lum = st.f64(it), st.f64(ig), st.f64(il)

    if av_table is not None:
        edges, mu, sig = av_table
        if len(edges) != len(mu) + 1 or len(mu) != len(sig):
            raise ValueError('av_table must be (edges[n+1], mu[n], sigma[n])')
        P.nav = len(mu)
        P.av_edges_pc, P.av_mu, P.av_sig = st.f64(edges), st.f64(mu), st.f64(sig)
    else:
        P.nav = 0
        P.av_edges_pc, P.av_mu, P.av_sig = st.f64([]), st.f64([]), st.f64([])
    P.tmin, P.tmax = float(tmin), float(tmax)
    mean, sig, has = prior_vectors(prior, nspec, dist_fit)
    for i in range(_lib.MAX_DIM):
        P.prior_mean[i] = mean[i]
        P.prior_sig[i] = sig[i]
    P.use_av = int(bool(use_av))
    P.dist_fit = int(bool(dist_fit))
    P.rad_prior = int(bool(rad_prior))
    P.has_prior_list = has
    P.no_spectrum = 0 if spectrum else 1  # mft6_nospec.py drops the spectrum term
    st.window = (j0, nwin)
    st.r, st.tmi, st.tma = [float(x) for x in r], float(tmi), float(tma)
    st.phot_cwl = np.array(phot_cwl)
    st.nc, st.nph = nc, nph
    return st


def isochrone_logg(teff, matrix):
    """Host-side logg(Teff) for building inputs (same table/ordering the device lookup uses)."""
    x, g, _ = sorted_isochrone(np.asarray(matrix))
    if np.any(np.asarray(teff) < x[0]) or np.any(np.asarray(teff) > x[-1]):
        raise ValueError('A value in x_new is outside the interpolation range (isochrone Teff)')
    return np.interp(teff, x, g)


def isochrone_radius(teff, matrix):
    """Host-side Stefan-Boltzmann radius [Rsun] from the isochrone luminosity (get_radius, mft6.py:66-85)."""
    x, _, lum = sorted_isochrone(np.asarray(matrix))
    if np.any(np.asarray(teff) < x[0]) or np.any(np.asarray(teff) > x[-1]):
        raise ValueError('A value in x_new is outside the interpolation range (isochrone Teff)')
    lm = np.interp(teff, x, lum)
    sigma_sb, lsun, rsun = 5.670374e-5, 3.839e33, 6.957e10
    return np.sqrt(lm * lsun / (4 * np.pi * sigma_sb * np.asarray(teff, dtype=float) ** 4)) / rsun
