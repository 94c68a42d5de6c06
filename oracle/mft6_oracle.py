"""CPU oracle: a NumPy/SciPy restatement of the reference's per-walker log-likelihood path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mcmc_spec_amd/`` may import this module; it exists so
that ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg can check / time
the HIP path against an independent float64 statement of what ``/root/reference/mft6.py`` computes.

Pinning status:
  * every numpy/scipy step (A1, A2, A4, A5, A8, A9, prior box/Gaussians) is pinned against the
    reference's *own functions* executed in the build container by ``tests/golden/make_golden.py``
    (stub-import of ``mft6.py``; outputs committed as ``tests/golden/*.npz``);
  * third-party arithmetic whose source is not in the container -- ``extinction.ccm89/apply``
    (A7), ``PyAstronomy.pyasl.instrBroadGaussFast`` (A3), ``pyphot`` band fluxes / zero points
    (A6), ``dustmaps`` Bayestar (prior) -- is restated from the published algorithms and is
    **parity unpinned** (the reference has no tests or vectors for them).

Each function cites the reference lines it follows.  The one deliberate departure: the reference
re-lists the model directory with ``glob`` on every ``get_spec`` call (mft6.py:423-436,457) to learn
the node values; here the node lists come from the keys of ``specs`` (or are passed explicitly).
"""
from __future__ import annotations

import numpy as np
from scipy.interpolate import interp1d

RSUN_CM = 6.957e10  # mft6.py:691
PC_CM = 3.086e18  # mft6.py:691


# --------------------------------------------------------------------------------------------- A1
def _iso_rows(matrix):
    """Rows of the 1 Gyr isochrone: ``matrix[:,1] == 9.0``, first 220 (mft6.py:90-95)."""
    sel = np.where(np.asarray(matrix[:, 1]) == 9.0)[0][:220]
    return sel


def get_logg(teff, matrix):
    """logg(Teff) by linear interpolation in the isochrone (mft6.py:87-98)."""
    sel = _iso_rows(matrix)
    return interp1d(matrix[sel, 4], matrix[sel, 5])(teff)


def get_radius(teff, matrix):
    """Stefan-Boltzmann radius in Rsun from the isochrone luminosity (mft6.py:66-85)."""
    sel = _iso_rows(matrix)
    lum = interp1d(matrix[sel, 4], matrix[sel, 6])(teff)
    sigma_sb, lsun = 5.670374e-5, 3.839e33
    return np.sqrt(lum * lsun / (4 * np.pi * sigma_sb * teff**4)) / RSUN_CM


# --------------------------------------------------------------------------------------------- A2
def nodes_from_specs(specs):
    """Sorted unique Teff (int) and logg (float) node values present as keys of ``specs``."""
    ts, gs = set(), set()
    for k in specs:
        if k == 'wl':
            continue
        a, b = k.split(', ')
        ts.add(int(a))
        gs.add(float(b))
    return sorted(ts), sorted(gs)


def bracket(nodes, value):
    """Nearest node first, then its neighbour on the other side of ``value`` (mft6.py:439-453,467-477).

    Returns (e1, e2).  e2 == e1 when ``value`` sits exactly on a node.  Index arithmetic is left
    unchecked exactly like the reference (below the lowest node wraps to the last one, above the
    highest raises IndexError).
    """
    arr = np.asarray(nodes)
    i1 = int(np.abs(arr - value).argmin())  # mft6.py:100-113, first index on ties
    if nodes[i1] == value:
        i2 = i1
    elif nodes[i1] > value:
        i2 = i1 - 1
    else:
        i2 = i1 + 1
    return nodes[i1], nodes[i2]


def lerp_spec(s1, s2, e1, e2, v):
    """mft6.py:205 -- ``(S2-S1)/(e2-e1) * (v-e1) + S1``."""
    return ((np.asarray(s2) - np.asarray(s1)) / (e2 - e1)) * (v - e1) + np.asarray(s1)


def get_spec(temp, log_g, reg, specs, temps=None, lgs=None):
    """Bilinear (logg first, then Teff) blend of pre-loaded nodes, cropped to ``reg`` [um] (mft6.py:387-563)."""
    if temps is None or lgs is None:
        t_all, g_all = nodes_from_specs(specs)
        temps = t_all if temps is None else temps
        lgs = g_all if lgs is None else lgs
    t1, t2 = bracket(temps, temp)
    g1, g2 = bracket(lgs, log_g)
    wave = specs['wl']
    key = '{}, {}'.format
    if g1 == g2 and t1 == t2:  # mft6.py:488-489
        flux = specs[key(t1, g1)]
    else:
        s11 = specs[key(t1, g1)]
        s22 = specs[key(t2, g2)]
        s12 = specs[key(t1, g2)]
        s21 = specs[key(t2, g1)]
        if g1 != g2 and t1 != t2:  # mft6.py:507-511
            a = lerp_spec(s11, s12, g1, g2, log_g)
            b = lerp_spec(s21, s22, g1, g2, log_g)
            flux = lerp_spec(a, b, t1, t2, temp)
        elif g1 == g2:  # mft6.py:514-515
            flux = lerp_spec(s11, s22, t1, t2, temp)
        else:  # mft6.py:518-519
            flux = lerp_spec(s11, s22, g1, g2, log_g)
    lim = np.array(reg) * 1e4  # mft6.py:537
    keep = np.where((wave >= min(lim)) & (wave <= max(lim)))  # mft6.py:542
    return np.array(wave[keep]), np.array(flux[keep])


# --------------------------------------------------------------------------------------------- A3
def broad_gauss_kernel(dx, sigma, maxsig=5):
    """Kernel of PyAstronomy ``broadGaussFast``: length ``int(sigma*maxsig/dx*2)+1``, offsets
    ``(arange(lx) - (lx//2 + lx%2) + 1)*dx``, Gaussian normalised to unit sum (published algorithm)."""
    lx = int(((sigma * maxsig) / dx) * 2.0) + 1
    nx = (np.arange(lx, dtype=int) - sum(divmod(lx, 2)) + 1) * dx
    e = 1.0 / np.sqrt(2.0 * np.pi * sigma**2) * np.exp(-(nx**2) / (2.0 * sigma**2))
    return e / np.sum(e)


def instr_broad_gauss_fast(wvl, flux, resolution, maxsig=5):
    """PyAstronomy ``pyasl.instrBroadGaussFast(wvl, flux, res, maxsig=5)`` as called at mft6.py:128."""
    wvl = np.asarray(wvl, dtype=float)
    dxs = wvl[1:] - wvl[:-1]
    if abs(max(dxs) - min(dxs)) > np.mean(dxs) * 1e-6:
        raise ValueError('The wavelength axis is not equidistant')
    fwhm = 1.0 / float(resolution) * np.mean(wvl)
    sigma = fwhm / (2.0 * np.sqrt(2.0 * np.log(2.0)))
    e = broad_gauss_kernel(dxs[0], sigma, maxsig)
    return np.convolve(np.asarray(flux, dtype=float), e, mode='same')


def broaden(even_wl, flux, res):
    """mft6.py:124-152 with vsini = limb = 0: broadened flux with the two edge patches (:129-130)."""
    b = instr_broad_gauss_fast(even_wl, flux, res, maxsig=5)
    b[0:5] = b[5]
    b[len(b) - 10:len(b)] = b[len(b) - 11]
    return np.array(even_wl), np.array(b)


def broaden_specs_window(specs_raw, w, resolution):
    """The per-node staging step of ``spec_interpolator`` (mft6.py:366-383): broaden only the data
    window ``[min(w), max(w)]`` [A] of every node and splice it back between the untouched wings."""
    wl = specs_raw['wl']
    inside = np.where((wl >= min(w)) & (wl <= max(w)))
    below, above = np.where(wl < min(w)), np.where(wl > max(w))
    out = {}
    for k, v in specs_raw.items():
        if k == 'wl':
            continue
        ww, brd = broaden(wl[inside], v[inside], resolution)
        out[k] = np.concatenate((v[below], brd, v[above]))
    out['wl'] = np.concatenate((wl[below], ww, wl[above]))
    return out


# --------------------------------------------------------------------------------------------- A6
class OracleBand:
    """One passband with the members of a pyphot ``Filter`` the reference touches
    (``get_flux``, ``Vega_zero_flux``, ``AB_zero_flux``; mft6.py:773-782).  Photon-counting type.
    Follows pyphot's published ``UnitFilter.get_flux`` / ``AB_zero_mag`` -- parity unpinned."""

    def __init__(self, wavelength_aa, transmit, vega_wl=None, vega_flux=None):
        self.wavelength = np.asarray(wavelength_aa, dtype=float)
        self.transmit = np.asarray(transmit, dtype=float)
        self._lT = np.trapz(self.wavelength * self.transmit, self.wavelength)
        self._lpivot2 = self._lT / np.trapz(self.transmit / self.wavelength, self.wavelength)
        c_aa_s = 2.99792458e18
        self.AB_zero_mag = 2.5 * np.log10(self._lpivot2 / c_aa_s) + 48.6
        self.AB_zero_flux = 10.0 ** (-0.4 * self.AB_zero_mag)
        self.Vega_zero_flux = None
        if vega_wl is not None:
            self.Vega_zero_flux = self.get_flux(vega_wl, vega_flux)

    def get_flux(self, slamb, sflux):
        slamb = np.asarray(slamb, dtype=float)
        sflux = np.asarray(sflux, dtype=float)
        ift = np.interp(slamb, self.wavelength, self.transmit, left=0.0, right=0.0)
        nz = np.where(ift > 0.0)[0]
        if nz.size <= 0:
            return 0.0
        a0 = max(0, min(nz) - 5)
        a1 = min(len(ift), max(nz) + 5)
        sl = slice(a0, a1)
        num = np.trapz(slamb[sl] * ift[sl] * sflux[sl], slamb[sl])
        den = np.trapz(slamb[sl] * ift[sl], slamb[sl])
        return num / den


def make_band_library(tables, vega_wl, vega_flux):
    """dict name -> OracleBand, the stand-in for ``pyphot.get_library()`` (mft6.py:21)."""
    return {k: OracleBand(w, t, vega_wl, vega_flux) for k, (w, t) in tables.items()}


PHOT_BANDS_3 = ['2MASS_J', '2MASS_H', '2MASS_Ks']  # mft6.py:767
PHOT_BANDS_6 = ['SDSS_r', 'SDSS_i', 'SDSS_z', '2MASS_J', '2MASS_H', '2MASS_Ks']  # mft6.py:769


# ------------------------------------------------------------------------------------------ A4-A6
def composite_window(r, tmi, tma, ctm, ptm):
    """The [lo, hi] window in micron that ``make_composite`` hands to ``get_spec`` (mft6.py:663-673,687)."""
    wlmin, wlmax = np.inf, 0
    for w in list(ctm[0]) + list(ptm[0]):
        wlmin = min(wlmin, min(w))
        wlmax = max(wlmax, max(w))
    return [min(min(r), tmi / 1e4, wlmin / 1e4) - 1e-4, max(max(r), tma / 1e4, wlmax / 1e4) + 1e-4]


def make_composite(teff, logg, rad, distance, contrast_filt, phot_filt, r, specs, ctm, ptm, tmi, tma,
                   nspec=2, bandlib=None, temps=None, lgs=None):
    """Composite spectrum, contrasts and unresolved photometry (mft6.py:651-831, ``plot=False``)."""
    wls, tras = ctm[0], ctm[1]
    phot_cwl = ptm[3]
    reg = composite_window(r, tmi, tma, ctm, ptm)
    stars = []
    wave = None
    for n in range(len(teff)):
        wave, s = get_spec(teff[n], logg[n], reg, specs, temps=temps, lgs=lgs)
        if not type(distance) == bool:
            di = 1 / distance
            if n == 0:
                s = s * (rad[0] * RSUN_CM / (di * PC_CM)) ** 2  # mft6.py:691
            else:
                s = s * (rad[0] * rad[n] * RSUN_CM / (di * PC_CM)) ** 2  # mft6.py:700
        elif n > 0:
            s = s * rad[n - 1] ** 2  # mft6.py:703
        stars.append(s)
    stars = np.vstack(stars)

    mag = np.zeros((len(contrast_filt), len(teff)))
    for n in range(len(contrast_filt)):  # mft6.py:717-733
        ran, tm = wls[n], tras[n]
        inband = np.where((wave <= max(ran)) & (wave >= min(ran)))
        w = wave[inband]
        tran = interp1d(ran, tm)(w)
        for k in range(len(teff)):
            m = np.trapz(stars[k][inband] * tran, w)
            mag[n][k] = -2.5 * np.log10(m)

    if float(nspec) == 2:  # mft6.py:740-744
        contrast = [mag[n][1] - mag[n][0] for n in range(len(contrast_filt))]
        comp = stars[0] + stars[1]
    else:  # mft6.py:746-751
        c1 = [mag[n][1] - mag[n][0] for n in range(len(contrast_filt))]
        c2 = [mag[n][2] - mag[n][0] for n in range(len(contrast_filt))]
        h = int(len(contrast_filt) / 2)
        contrast = list(np.concatenate((c1[:h], c2[h:])))
        comp = stars[0] + stars[1] + stars[2]

    names = PHOT_BANDS_3 if len(phot_filt) == 3 else PHOT_BANDS_6  # mft6.py:766-769
    phot = []
    for n in range(len(phot_filt)):  # mft6.py:771-783
        band = bandlib[names[n]]
        f = band.get_flux(wave, comp)
        zero = band.Vega_zero_flux if '2MASS' in names[n] else band.AB_zero_flux
        phot.append(-2.5 * np.log10(f / zero))
    return (np.array(wave), np.array(comp), [c for c in contrast],
            np.array([float(p) for p in phot_cwl]), np.array(phot), stars)


# --------------------------------------------------------------------------------------------- A7
def ccm89_ab(x):
    """Cardelli, Clayton & Mathis (1989) a(x), b(x); x in inverse micron (published coefficients;
    what ``extinction.ccm89`` evaluates -- source not in the container, parity unpinned)."""
    x = np.atleast_1d(np.asarray(x, dtype=float))
    a = np.empty_like(x)
    b = np.empty_like(x)
    ir = x < 1.1
    opt = (x >= 1.1) & (x < 3.3)
    uv = (x >= 3.3) & (x < 8.0)
    fuv = x >= 8.0
    y = x[ir] ** 1.61
    a[ir], b[ir] = 0.574 * y, -0.527 * y
    y = x[opt] - 1.82
    a[opt] = ((((((0.329990 * y - 0.77530) * y + 0.01979) * y + 0.72085) * y - 0.02427) * y - 0.50447) * y
              + 0.17699) * y + 1.0
    b[opt] = ((((((-2.09002 * y + 5.30260) * y - 0.62251) * y - 5.38434) * y + 1.07233) * y + 2.28305) * y
              + 1.41338) * y
    xu = x[uv]
    au = 1.752 - 0.316 * xu - 0.104 / ((xu - 4.67) ** 2 + 0.341)
    bu = -3.090 + 1.825 * xu + 1.206 / ((xu - 4.62) ** 2 + 0.263)
    yb = np.clip(xu - 5.9, 0.0, None)
    au += -0.04473 * yb**2 - 0.009779 * yb**3
    bu += 0.2130 * yb**2 + 0.1207 * yb**3
    a[uv], b[uv] = au, bu
    y = x[fuv] - 8.0
    a[fuv] = -0.070 * y**3 + 0.137 * y**2 - 0.628 * y - 1.073
    b[fuv] = 0.374 * y**3 - 0.420 * y**2 + 4.257 * y + 13.670
    return a, b


def ccm89(wave_aa, a_v, r_v=3.1):
    """``extinction.ccm89(wave, a_v, r_v)``: A(lambda) in magnitudes, wavelengths in Angstrom."""
    a, b = ccm89_ab(1e4 / np.asarray(wave_aa, dtype=float))
    return a_v * (a + b / r_v)


def extinct(wl, spec, av, rv=3.1):
    """mft6.py:46-64: ``extinction.apply(ccm89(wl, av, rv), spec)`` = spec * 10^(-0.4 A)."""
    return np.array(np.asarray(spec) * 10.0 ** (-0.4 * ccm89(wl, av, rv)))


# ------------------------------------------------------------------------------------------ A8-A9
def chisq(model, data, var):
    """mft6.py:115-122 (``var`` is a sigma; squared inside)."""
    return ((np.array(model) - np.array(data)) ** 2) / np.array(var) ** 2


def norm_spec(wl, model, data):
    """mft6.py:193-196: divide ``data`` by the quadratic least-squares fit of data/model vs wl."""
    p = np.polynomial.Polynomial.fit(wl, data / model, deg=2)
    return data / p(wl)


def loglikelihood(p0, fr, nspec, data, err, r, specs, ctm, ptm, tmi, tma, matrix, av=True, optimize=False,
                  bandlib=None, temps=None, lgs=None, parts=None, spectrum=True, inpath=None):
    """mft6.py:1139-1205 for len(p0) in (6, 8).  ``parts`` (a dict) receives intermediates for tests.

    ``inpath=dict(specs_raw=..., w=[wmin, wmax] (Angstrom), resolution=R)``: SURVEY A3 placement (ii) -- the composite of the
    spectrum term is built from the UNbroadened node spectra and ``broaden`` (mft6.py:124-152, the call the reference keeps
    commented out at :550, restricted here to the data window like the staging step :373) is applied to it per evaluation,
    before the reddening; ``specs`` (broadened per node at staging) still feeds the contrast and photometry terms."""
    wl, spec = np.array(data)
    t_guess = p0[:nspec]
    a_v = p0[nspec]
    rad = p0[nspec + 1:2 * nspec + 1]
    plx = p0[2 * nspec + 1]
    lg = [get_logg(t, matrix) for t in t_guess]  # mft6.py:1149
    wave1, cspec, contrast, phot_cwl, phot, _ = make_composite(
        t_guess, lg, rad, plx, fr[2], fr[5], r, specs, ctm, ptm, tmi, tma, nspec=nspec, bandlib=bandlib,
        temps=temps, lgs=lgs)
    if inpath is not None:
        wave_r, cspec_r = make_composite(t_guess, lg, rad, plx, fr[2], fr[5], r, inpath['specs_raw'], ctm, ptm, tmi, tma,
                                         nspec=nspec, bandlib=bandlib, temps=temps, lgs=lgs)[:2]
        inside = np.where((wave_r >= min(inpath['w'])) & (wave_r <= max(inpath['w'])))
        _, brd = broaden(wave_r[inside], cspec_r[inside], inpath['resolution'])
        cspec = np.array(cspec_r, dtype=float)
        cspec[inside] = brd
        wave1 = wave_r
    if av == True and a_v > 0:  # mft6.py:1161-1163
        cspec = extinct(wave1, cspec, a_v)
        init_phot = -2.5 * np.log10(extinct(phot_cwl, 10 ** (-0.4 * phot), a_v))
    else:
        init_phot = phot
    model = interp1d(wave1, cspec)(wl * 1e4)  # mft6.py:1169-1170
    model = model * (np.median(spec) / np.median(model))  # mft6.py:1173
    spec_n = norm_spec(wl, model, spec)  # mft6.py:1174
    ic = chisq(model, spec_n, err)
    iic = np.sum(ic) / len(ic)  # mft6.py:1179
    chi_c = chisq(contrast, fr[0], fr[1])
    chi_p = chisq(init_phot, fr[3], fr[4])
    total = np.sum((iic * (len(chi_c) + len(chi_p)), np.sum(chi_c), np.sum(chi_p)))  # mft6.py:1191
    if not spectrum:  # mft6_nospec.py:1192: the spectrum term is commented out there
        total = np.sum((np.sum(chi_c), np.sum(chi_p)))
    if parts is not None:
        parts.update(logg=np.array(lg, dtype=float), contrast=np.array(contrast), phot=np.array(init_phot),
                     model=model, data_norm=spec_n, iic=iic, icontrast=np.sum(chi_c), iphot=np.sum(chi_p))
    if optimize:
        return total
    return -np.inf if np.isnan(total) else -0.5 * total


# ------------------------------------------------------------------------------- adjacent: prior
def logprior(p0, nspec, tmin, tmax, matrix, av_prior, prior=0, ext=True, dist_fit=True, rad_prior=False):
    """mft6.py:1207-1272 (len 6, dist_fit) and :1329-1393 (len 8, dist_fit).

    ``av_prior(distance_pc) -> (mu, sigma)`` stands in for
    ``bayestar(SkyCoord(ra, dec, 1/plx pc), mode='samples') * 3.1 * 0.884`` mean/std (mft6.py:1233-1238);
    sigma == 0 is replaced by 0.05 as in the reference.
    """
    if not dist_fit:
        return _logprior_no_dist(p0, nspec, tmin, tmax, matrix, av_prior, prior, ext, rad_prior)
    temps = p0[:nspec]
    a_v = p0[nspec]
    rad = list(p0[nspec + 1:2 * nspec + 1])
    dist = p0[2 * nspec + 1]
    if len(p0) == 6:
        if (any(t > tmax for t in temps) or any(t < tmin for t in temps) or any(x < 0.05 for x in rad)
                or rad[0] > 1.5 or dist < 1 / 3000 or dist > 1 / 4):  # mft6.py:1227
            return -np.inf
    else:
        if (any(t > tmax for t in temps) or any(t < tmin for t in temps) or any(x < 0.05 for x in rad)
                or dist < 1 / 1000 or dist > 1 / 4):  # mft6.py:1347
            return -np.inf
    pp = []
    if ext:
        if a_v < 0:  # mft6.py:1229
            return -np.inf
        mu, sig = av_prior(1.0 / dist)
        if sig == 0:
            sig = 0.05
        pp.append(-0.5 * ((a_v - mu) / sig) ** 2)
    if not (isinstance(prior, (int, float)) and prior == 0):  # mft6.py:1241-1260
        prior = list(prior)
        ps = prior[:nspec] + [prior[2 * nspec]] + prior[2 * nspec + 2:3 * nspec + 2] + [prior[-2]]
        ss = prior[nspec:2 * nspec] + [prior[2 * nspec + 1]] + prior[3 * nspec + 2:4 * nspec + 2] + [prior[-1]]
        for k, p in enumerate(ps):
            if p != 0:
                pp.append(-0.5 * ((p0[k] - p) / ss[k]) ** 2)
    if rad_prior:  # mft6.py:1262-1269 / 1383-1390
        mr = [get_radius(t, matrix) for t in temps]
        targets = [mr[0]] + [m / mr[0] for m in mr[1:]]
        for k, p in enumerate(targets):
            pp.append(-0.5 * ((rad[k] - p) / (0.02 * p)) ** 2)
    return np.sum(pp)


def _logprior_no_dist(p0, nspec, tmin, tmax, matrix, av_prior, prior, ext, rad_prior):
    """``dist_fit=False``: mft6.py:1275-1327 (len 6) and :1397-1454 (len 8).  No parallax / absolute-radius
    bounds, a shorter Gaussian-prior list, the A_V prior still evaluated at distance 1/p0[-1]."""
    temps = p0[:nspec]
    if len(p0) == 6:
        rad, rad1 = p0[-2], p0[-3]
        a_v = p0[-4]
        if any(t > tmax for t in temps) or any(t < tmin for t in temps) or rad < 0.05 or rad1 < 0.05:  # :1286
            return -np.inf
        radii = [rad1, rad]
    else:
        rad1, rad2, rad, dist = p0[-3], p0[-2], p0[-4], p0[-1]
        a_v = p0[-5]
        if (any(t > tmax for t in temps) or any(t < tmin for t in temps) or rad1 < 0.05 or rad2 < 0.05
                or dist < 0):  # :1411
            return -np.inf
        radii = [rad, rad1, rad2]
    pp = []
    if ext:
        if a_v < 0:
            return -np.inf
        mu, sig = av_prior(1.0 / p0[-1])
        if sig == 0:
            sig = 0.05
        pp.append(-0.5 * ((a_v - mu) / sig) ** 2)
    if not (isinstance(prior, (int, float)) and prior == 0):  # mft6.py:1301-1318 / :1425-1442
        prior = list(prior)
        ps = prior[:nspec] + [prior[2 * nspec]] + prior[2 * nspec + 2:3 * nspec + 1]
        ss = prior[nspec:2 * nspec] + [prior[2 * nspec + 1]] + prior[3 * nspec + 2:4 * nspec + 1]
        for k, p in enumerate(ps):
            if p != 0:
                pp.append(-0.5 * ((p0[k] - p) / ss[k]) ** 2)
    if rad_prior:
        mr = [get_radius(t, matrix) for t in temps]
        targets = [mr[0]] + [m / mr[0] for m in mr[1:]]
        for k, p in enumerate(targets):
            pp.append(-0.5 * ((radii[k] - p) / (0.02 * p)) ** 2)
    return np.sum(pp)


def logposterior(p0, fr, nspec, data, err, r, specs, ctm, ptm, tmi, tma, tmin, tmax, matrix, av_prior,
                 prior=0, a=True, dist_fit=True, rad_prior=False, bandlib=None, temps=None, lgs=None, inpath=None):
    """mft6.py:1459-1470.  (``inpath``: see loglikelihood.)"""
    lp = logprior(p0, nspec, tmin, tmax, matrix, av_prior, prior=prior, ext=a, dist_fit=dist_fit,
                  rad_prior=rad_prior)
    if not np.isfinite(lp):
        return -np.inf
    lh = loglikelihood(p0, fr, nspec, data, err, r, specs, ctm, ptm, tmi, tma, matrix, av=a,
                       bandlib=bandlib, temps=temps, lgs=lgs, inpath=inpath)
    return lp + lh


# ------------------------------------------------------------------------------------- f3: loader
def spec_interpolator(w, trange, lgrange, specrange, resolution=10000, grid_dir='BT-Settl_M-0.0a+0.0'):
    """``spec_interpolator(..., models='btsettl')`` (mft6.py:323-385): read two-column BT-Settl text
    files ``lte{TTT}-{g}-0.0a+0.0.BT-Settl.spec.7.txt``, keep samples within +-100 A of ``specrange``,
    linearly resample every node onto ``np.arange(min, max, 0.2)``, broaden the data window
    ``[min(w), max(w)]`` and splice.  ``w`` and ``specrange`` in Angstrom."""
    from glob import glob
    import os
    files = glob(os.path.join(grid_dir, 'lte*'))
    t, l = [], []
    for f in files:  # mft6.py:330-340
        base = os.path.basename(f)
        nu = int(float(base.split('-')[0].split('e')[1]) * 1e2)
        mu = float(base.split('-')[1])
        if nu not in t and min(trange) <= nu <= max(trange):
            t.append(nu)
        if mu not in l and min(lgrange) <= mu <= max(lgrange):
            l.append(mu)
    wl = np.arange(min(specrange), max(specrange), 0.2)  # mft6.py:343
    raw = {}
    for tt in t:
        for ll in l:
            name = os.path.join(grid_dir, 'lte{}-{}-0.0a+0.0.BT-Settl.spec.7.txt'.format(str(int(tt / 1e2)).zfill(3), str(ll)))
            xs, ys = [], []
            with open(name) as fh:  # mft6.py:353-357
                for line in fh:
                    li = line.split(' ')
                    if min(specrange) - 100 <= float(li[0]) <= max(specrange) + 100:
                        xs.append(float(li[0]))
                        ys.append(float(li[1]))
            raw['{}, {}'.format(tt, ll)] = interp1d(np.array(xs), np.array(ys))(wl)  # mft6.py:369-371
    raw['wl'] = wl
    return broaden_specs_window(raw, w, resolution)  # mft6.py:373-383


# ------------------------------------------------------------------------ f4: pre-optimiser chi^2
def opt_prior(vals, pval, psig):
    """mft6.py:833-854 (chi^2-form Gaussian terms; entries with pval == 0 are skipped in the list branch)."""
    pp = []
    if len(pval) == 1 or type(pval) == float:
        try:
            pp.append(((float(vals) - float(pval)) / float(psig)) ** 2)
        except Exception:
            pp.append(((vals[0] - pval[0]) / psig[0]) ** 2)
    else:
        for k, p in enumerate(pval):
            if p != 0:
                pp.append(((vals[k] - pval[k]) / (psig[k])) ** 2)
    return np.sum(pp)


def fit_spec_init(wl_aa, flux, err, reg, t_guess, rad_guess, plx, fr, specs, ctm, ptm, tmi, tma, matrix, nspec=2,
                  bandlib=None):
    """The initial guess of ``fit_spec`` (mft6.py:871-907): un-reddened composite, data normalised ONCE
    against it.  Returns (likelihood chi^2 with spectrum weight 3, normalised data vector)."""
    lg = [get_logg(t, matrix) for t in t_guess]
    wave1, cspec, contrast, phot_cwl, phot, _ = make_composite(t_guess, lg, rad_guess, plx, fr[2], fr[5], reg, specs,
                                                               ctm, ptm, tmi, tma, nspec=nspec, bandlib=bandlib)
    model = interp1d(wave1, cspec)(wl_aa)  # mft6.py:884-885 (no extinct: :880 is commented out)
    model = model * (np.median(flux) / np.median(model))  # mft6.py:888
    flux_n = norm_spec(wl_aa, model, flux)  # mft6.py:889
    iic = np.sum(chisq(model, flux_n, err)) / len(flux_n) * 3  # mft6.py:892-893
    chi_c = chisq(contrast, fr[0], fr[1])
    ip = chisq(phot, fr[3], fr[4])  # mft6.py:901 uses the UN-reddened photometry
    return np.sum((iic * (len(chi_c) + len(ip)), np.sum(chi_c), np.sum(ip))), flux_n


def fit_spec_proposal(wl_aa, flux_n, err, reg, teff, a_v, rad, plx, fr, specs, ctm, ptm, tmi, tma, matrix, nspec=2,
                      bandlib=None):
    """Likelihood chi^2 of one proposal inside ``fit_spec`` (mft6.py:997-1028): reddened if A_V > 0,
    median-scaled to the already-normalised data, NO per-proposal continuum fit, spectrum weight 3."""
    lg = [get_logg(v, matrix) for v in teff]
    wave1, cspec, contrast, phot_cwl, phot, _ = make_composite(teff, lg, rad, float(plx), fr[2], fr[5], reg, specs, ctm,
                                                               ptm, tmi, tma, nspec=nspec, bandlib=bandlib)
    if a_v > 0:  # mft6.py:1002-1004
        cspec = extinct(wave1, cspec, a_v)
        phot = -2.5 * np.log10(extinct(phot_cwl, 10 ** (-0.4 * phot), a_v))
    model = interp1d(wave1, cspec)(wl_aa)
    model = model * (np.median(flux_n) / np.median(model))  # mft6.py:1011
    ttc = np.sum(chisq(model, flux_n, err)) / len(flux_n) * 3  # mft6.py:1014-1015
    chi_c = chisq(contrast, fr[0], fr[1])
    chi_p = chisq(phot, fr[3], fr[4])
    return np.sum((ttc * (len(chi_c) + len(chi_p)), np.sum(chi_c), np.sum(chi_p)))  # mft6.py:1028
