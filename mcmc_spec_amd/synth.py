"""Deterministic synthetic stand-ins for the model data the reference needs but does not ship.

The reference reads a BT-Settl text grid (``mft6.py:323-383``), a MIST isochrone table
(``mft6.py:3487-3490``), filter curves from ``bps/`` (``mft6.py:565-649``), pyphot's bundled
passbands + Vega spectrum (``mft6.py:21,773-782``) and the Bayestar dust map (``mft6.py:23,1233``).
None of the large ones are in the repository, so the benchmarks, the smoke run and
the GPU parity tests regenerate *shape-compatible* stand-ins from fixed seeds.  Everything here is
labelled synthetic; nothing claims astrophysical truth.

All generators are pure NumPy and only use +,-,*,/ and exp so that two machines produce the same
arrays to ~1 ulp.
"""
from __future__ import annotations

import numpy as np

H_CGS = 6.62607015e-27
C_CGS = 2.99792458e10
K_CGS = 1.380649e-16

DEFAULT_SEED = 20241220  # reference snapshot date, SURVEY.md §8(d)


def planck_surface_flux(wl_aa: np.ndarray, teff: float) -> np.ndarray:
    """pi * B_lambda(T) in erg s^-1 cm^-2 A^-1 per unit stellar surface (the BT-Settl unit, mft6.py:686)."""
    lam = wl_aa * 1e-8
    x = H_CGS * C_CGS / (lam * K_CGS * teff)
    b = 2.0 * H_CGS * C_CGS**2 / lam**5 / np.expm1(x)
    return np.pi * b * 1e-8


class LineForest:
    """A seeded absorption-line forest on a fixed wavelength grid, shared by every node.

    Each line i has a Gaussian profile (centre, width) and a node-dependent strength
    ``d_i * max(0, 1 + a_i*(T-4000)/1000 + b_i*(g-4.75))`` so that bilinear interpolation between
    nodes is not trivially exact.  Stored sparsely: (flat grid index, line id, profile value).
    """

    def __init__(self, wl: np.ndarray, nlines: int = 4000, seed: int = DEFAULT_SEED):
        rng = np.random.default_rng(seed)
        lo, hi = float(wl[0]), float(wl[-1])
        self.centre = rng.uniform(lo, hi, nlines)
        self.depth = rng.uniform(0.0, 0.6, nlines)
        self.width = rng.uniform(0.3, 2.0, nlines)
        self.at = rng.uniform(-0.8, 0.8, nlines)
        self.ag = rng.uniform(-0.8, 0.8, nlines)
        step = float(wl[1] - wl[0])
        idx_parts, line_parts, val_parts = [], [], []
        for i in range(nlines):
            half = int(5.0 * self.width[i] / step) + 1
            c = int(round((self.centre[i] - lo) / step))
            a, b = max(0, c - half), min(len(wl), c + half + 1)
            if b <= a:
                continue
            z = (wl[a:b] - self.centre[i]) / self.width[i]
            idx_parts.append(np.arange(a, b))
            line_parts.append(np.full(b - a, i))
            val_parts.append(np.exp(-0.5 * z * z))
        self.idx = np.concatenate(idx_parts)
        self.line = np.concatenate(line_parts)
        self.val = np.concatenate(val_parts)
        self.n = len(wl)
        # broad pseudo-molecular bands (period ~ 600-2500 A) that change with Teff
        self.band_period = rng.uniform(600.0, 2500.0, 6)
        self.band_phase = rng.uniform(0.0, 2 * np.pi, 6)
        self.band_amp = rng.uniform(0.02, 0.10, 6)

    def tau(self, teff: float, logg: float) -> np.ndarray:
        s = self.depth * np.maximum(0.0, 1.0 + self.at * (teff - 4000.0) / 1000.0 + self.ag * (logg - 4.75))
        return np.bincount(self.idx, weights=self.val * s[self.line], minlength=self.n)

    def bands(self, wl: np.ndarray, teff: float) -> np.ndarray:
        out = np.zeros_like(wl)
        cool = (5600.0 - teff) / 2600.0
        for p, ph, a in zip(self.band_period, self.band_phase, self.band_amp):
            out += a * cool * (1.0 + np.sin(2 * np.pi * wl / p + ph))
        return out


def make_grid(teff_nodes, logg_nodes, wl: np.ndarray, nlines: int = 4000, seed: int = DEFAULT_SEED) -> np.ndarray:
    """Synthetic model grid, float64 ``[nt][ng][nwl]``, strictly positive.

    flux(T, g, lambda) = pi*B_lambda(T) * exp(-tau_lines(T,g,lambda) - bands(T,lambda))
    """
    forest = LineForest(wl, nlines=nlines, seed=seed)
    teff_nodes = np.asarray(teff_nodes, dtype=float)
    logg_nodes = np.asarray(logg_nodes, dtype=float)
    out = np.empty((len(teff_nodes), len(logg_nodes), len(wl)))
    for it, t in enumerate(teff_nodes):
        cont = planck_surface_flux(wl, float(t))
        bands = forest.bands(wl, float(t))
        for ig, g in enumerate(logg_nodes):
            out[it, ig] = cont * np.exp(-(forest.tau(float(t), float(g)) + bands))
    return out


def grid_to_specs(teff_nodes, logg_nodes, wl, flux) -> dict:
    """Pack a dense grid into the reference's ``specs`` dict (keys ``'{T}, {g}'`` and ``'wl'``, mft6.py:363,383)."""
    specs = {}
    for it, t in enumerate(teff_nodes):
        for ig, g in enumerate(logg_nodes):
            specs['{}, {}'.format(int(t), float(g))] = flux[it, ig]
    specs['wl'] = np.asarray(wl)
    return specs


def make_isochrone_matrix(nrows: int = 220, extra_rows: int = 40) -> np.ndarray:
    """Stand-in for ``mist_2mass_old.cmd`` after the exponentiation at mft6.py:3489-3490.

    Columns used by the reference: 1 = log age (rows with exactly 9.0 are the 1 Gyr isochrone),
    4 = Teff [K], 5 = logg, 6 = L/Lsun.  The first ``nrows`` age-9 rows span 2900..6500 K with
    logg = 5.2 - 0.9*((T-2900)/3600)^1.3 (monotone).  Rows of other ages are interleaved so that the
    ``aage == 9.0`` filter (mft6.py:73,92) is exercised.
    """
    t = np.linspace(2900.0, 6500.0, nrows)
    x = (t - 2900.0) / 3600.0
    logg = 5.2 - 0.9 * x**1.3
    lum = 10.0 ** (-3.1 + 3.3 * x**0.8)
    main = np.zeros((nrows, 8))
    main[:, 0] = np.arange(nrows)
    main[:, 1] = 9.0
    main[:, 2] = 0.08 + 1.2 * x
    main[:, 3] = 0.0
    main[:, 4] = t
    main[:, 5] = logg
    main[:, 6] = lum
    other = np.zeros((extra_rows, 8))
    other[:, 1] = 8.5
    other[:, 4] = np.linspace(2800.0, 7000.0, extra_rows)
    other[:, 5] = 4.0
    other[:, 6] = 1.0
    tail = main[: min(30, nrows)].copy()  # age-9 rows beyond index 220 must be ignored ([:220])
    tail[:, 4] = np.linspace(6600.0, 9000.0, len(tail))
    tail[:, 5] = 4.0
    return np.vstack([other[: extra_rows // 2], main, tail, other[extra_rows // 2:]])


def make_av_table(nbins: int = 120, dmin_pc: float = 4.0, dmax_pc: float = 3000.0):
    """Stand-in for the Bayestar line of sight: (distance bin edges [pc], mu[bin], sigma[bin]).

    mu = 0.1 + 0.001*bin, sigma = 0.05.
    """
    edges = np.linspace(dmin_pc, dmax_pc, nbins + 1)
    mu = 0.1 + 0.001 * np.arange(nbins)
    sig = np.full(nbins, 0.05)
    return edges, mu, sig


def _bump(wl, lo, hi, soft):
    """Smooth top-hat between lo and hi with cosine edges of width ``soft``; zero at the end points."""
    y = np.ones_like(wl)
    a = wl < lo + soft
    y[a] = 0.5 - 0.5 * np.cos(np.pi * (wl[a] - lo) / soft)
    b = wl > hi - soft
    y[b] = 0.5 - 0.5 * np.cos(np.pi * (hi - wl[b]) / soft)
    return np.clip(y, 0.0, None)


def synthetic_contrast_filters():
    """Two contrast passbands shaped like the example's ``['lp600','Kp']`` (param_koi2298.txt:29).

    Returned in the reference's ``ctm`` layout ``[wls, tras, n_res_el, cwl]`` (mft6.py:3597) with
    wavelengths in Angstrom.  The first table is deliberately *unsorted with a repeated wavelength*
    (the real lp600/keck_kp tables are) so interp1d's sort semantics are exercised.
    """
    rng = np.random.default_rng(DEFAULT_SEED + 1)
    w1 = np.linspace(5976.0, 9954.0, 155)
    t1 = 0.9 * _bump(w1, 5976.0, 9954.0, 900.0) * (1.0 - 0.2 * (w1 - 5976.0) / 3978.0)
    perm = np.arange(len(w1))
    for k in rng.choice(len(w1) - 1, 12, replace=False):
        perm[k], perm[k + 1] = perm[k + 1], perm[k]
    w1, t1 = w1[perm], t1[perm]
    w2 = np.linspace(19100.0, 23325.7576, 65)
    t2 = 0.85 * _bump(w2, 19100.0, 23325.7576, 500.0)
    w2[12] = w2[11]  # a repeated abscissa, like keck_kp.txt rows 11-12
    wls = [list(w1), list(w2)]
    tras = [list(t1), list(t2)]
    res = 1700.0
    n_res = [(max(w) - min(w)) / (np.mean(w) / res) for w in wls]
    cwl = [float(np.mean(w)) for w in wls]
    return [wls, tras, n_res, cwl]


_BANDS = {
    # name: (lo, hi, soft edge, peak) in Angstrom; widths follow the SVO numbers quoted at mft6.py:759-760
    'SDSS_r': (5400.0, 7000.0, 300.0, 0.55),
    'SDSS_i': (6700.0, 8400.0, 300.0, 0.45),
    'SDSS_z': (7900.0, 10800.0, 500.0, 0.15),
    '2MASS_J': (10800.0, 14100.0, 500.0, 0.95),
    '2MASS_H': (14800.0, 18300.0, 500.0, 0.98),
    '2MASS_Ks': (19500.0, 23600.0, 600.0, 0.97),
}


def synthetic_band_tables(step: float = 25.0) -> dict:
    """Photon-counting passbands named like pyphot's library entries used at mft6.py:766-769."""
    out = {}
    for name, (lo, hi, soft, peak) in _BANDS.items():
        w = np.arange(lo, hi + 0.5 * step, step)
        out[name] = (w, peak * _bump(w, lo, hi, soft))
    return out


def synthetic_vega(step: float = 5.0):
    """A 9600 K blackbody scaled to 3.44e-9 erg/s/cm^2/A at 5556 A: stand-in for pyphot's Vega spectrum."""
    w = np.arange(2500.0, 32000.0, step)
    f = planck_surface_flux(w, 9600.0)
    f *= 3.44e-9 / np.interp(5556.0, w, f)
    return w, f


def synthetic_phot_filters():
    """``ptm`` for ``pfilt ['sdss,r','sdss,i','sdss,z','j','h','k']`` (param_koi2298.txt:33).

    In the reference ``ptm`` only contributes window extrema and ``phot_cwl``.
    """
    tabs = synthetic_band_tables()
    wls = [list(tabs[k][0]) for k in _BANDS]
    tras = [list(tabs[k][1]) for k in _BANDS]
    res = 1700.0
    n_res = [(max(w) - min(w)) / (np.mean(w) / res) for w in wls]
    cwl = [float(np.mean(w)) for w in wls]
    return [wls, tras, n_res, cwl]


def data_wavelengths_um(npix: int) -> np.ndarray:
    """Data pixel grid of the reference's synth files, in micron."""
    return (5500.684919966305 + 0.8424599831513557 * np.arange(npix)) / 1e4


TRUTH_THETA = np.array([3850.0, 3025.0, 0.106, 0.4994, 0.1546 / 0.4994, 2.0732e-3])
WALKER_SIGMA = np.array([50.0, 50.0, 0.02, 0.02, 0.02, 2e-5])


def draw_walkers(n: int, seed: int = 3, tmin: float = 3000.0, tmax: float = 5500.0) -> np.ndarray:
    """theta* + N(0, sigma), clipped into the prior box of mft6.py:1227."""
    rng = np.random.default_rng(seed)
    th = TRUTH_THETA + rng.normal(size=(n, 6)) * WALKER_SIGMA
    th[:, 0:2] = np.clip(th[:, 0:2], tmin + 1e-3, tmax - 1e-3)
    th[:, 2] = np.clip(th[:, 2], 1e-4, None)
    th[:, 3] = np.clip(th[:, 3], 0.05, 1.5)
    th[:, 4] = np.clip(th[:, 4], 0.05, None)
    th[:, 5] = np.clip(th[:, 5], 1 / 3000.0, 1 / 4.0)
    return th


EXAMPLE_CMAG = [2.08, 1.3]  # param_koi2298.txt:27
EXAMPLE_CERR = [0.14, 0.02]  # param_koi2298.txt:28
EXAMPLE_PMAG = [13.815, 13.505, 13.355, 12.323, 11.826, 11.735]  # param_koi2298.txt:31
EXAMPLE_PERR = [0.1, 0.1, 0.1, 0.026, 0.022, 0.019]  # param_koi2298.txt:32


def write_btsettl_text_grid(grid_dir, teffs=(3000, 3100, 3200, 3300), loggs=(4.5, 5.0), lo=4850.0, hi=9150.0,
                            seed=21):
    """Write BT-Settl-format text files ``lte{TTT}-{g}-0.0a+0.0.BT-Settl.spec.7.txt`` (mft6.py:251): two
    space-separated columns, wavelength [A] on an irregular grid (steps 0.05-0.35 A), flux.  Existing
    files in ``grid_dir`` are overwritten.  Returns ``grid_dir``."""
    import os
    os.makedirs(grid_dir, exist_ok=True)
    rng = np.random.default_rng(seed)
    for t in teffs:
        for g in loggs:
            steps = rng.uniform(0.05, 0.35, int((hi - lo) / 0.2) + 200)
            x = lo + np.cumsum(steps)
            x = x[x < hi]
            y = planck_surface_flux(x, float(t)) * (1.0 + 0.3 * np.sin(x / (3.0 + g)) * np.cos(x / 41.0))
            name = 'lte{}-{}-0.0a+0.0.BT-Settl.spec.7.txt'.format(str(int(t / 1e2)).zfill(3), str(float(g)))
            with open(os.path.join(grid_dir, name), 'w') as fh:
                for a, b in zip(x, y):
                    fh.write('{:.6f} {:.10e}\n'.format(a, b))
    return grid_dir
