"""Photometric passbands for the unresolved-photometry term.

The reference takes its bands from pyphot's bundled library (``lib = pyphot.get_library()``,
mft6.py:21; ``lib['2MASS_J']`` ..., mft6.py:766-773), integrates the composite with
``Filter.get_flux`` (photon-counting mean flux) and converts with ``Vega_zero_flux`` (2MASS) or
``AB_zero_flux`` (SDSS) (mft6.py:776-782).  pyphot is not installed here, so a band is the small
table below; ``from_pyphot`` adapts a real pyphot library when one is importable.

Only *static* per-band quantities are computed on the host here (integration weights on the model
wavelength grid, zero points); everything that touches model fluxes runs in the HIP library.
"""
from __future__ import annotations

import numpy as np

BAND_NAMES_3 = ['2MASS_J', '2MASS_H', '2MASS_Ks']  # len(phot_filt) == 3, mft6.py:766-767
BAND_NAMES_6 = ['SDSS_r', 'SDSS_i', 'SDSS_z', '2MASS_J', '2MASS_H', '2MASS_Ks']  # mft6.py:769

C_AA_PER_S = 2.99792458e18


def _trapz_weights(x):
    """w such that sum(w*y) == trapz(y, x)."""
    d = np.diff(x)
    w = np.zeros_like(x)
    w[:-1] += d / 2.0
    w[1:] += d / 2.0
    return w


class Band:
    """A photon-counting passband: wavelength [A], transmission, zero-point flux [erg/s/cm^2/A]."""

    def __init__(self, name, wavelength, transmit, zero_flux):
        self.name = name
        self.wavelength = np.asarray(wavelength, dtype=float)
        self.transmit = np.asarray(transmit, dtype=float)
        self.zero_flux = float(zero_flux)

    def weights_on(self, wave):
        """(i0, w) with ``sum(w * flux[i0:i0+len(w)])`` = pyphot's ``get_flux(wave, flux)``:
        T interpolated onto ``wave`` (0 outside), support padded by 5 samples, a/b with
        a = trapz(wave*T*flux), b = trapz(wave*T)."""
        ift = np.interp(wave, self.wavelength, self.transmit, left=0.0, right=0.0)
        nz = np.where(ift > 0.0)[0]
        if nz.size == 0:
            raise ValueError('band {} does not overlap the model window'.format(self.name))
        a0 = max(0, nz.min() - 5)
        a1 = min(len(ift), nz.max() + 5)
        x = wave[a0:a1]
        lt = x * ift[a0:a1]
        tw = _trapz_weights(x)
        den = np.sum(tw * lt)
        return int(a0), tw * lt / den

    def mean_flux(self, wave, flux):
        i0, w = self.weights_on(np.asarray(wave, dtype=float))
        return float(np.sum(w * np.asarray(flux, dtype=float)[i0:i0 + len(w)]))


def ab_zero_flux(wavelength, transmit):
    """pyphot ``AB_zero_flux`` of a photon-counting filter: 10^(-0.4*(2.5 log10(lpivot^2/c) + 48.6))."""
    w, t = np.asarray(wavelength, dtype=float), np.asarray(transmit, dtype=float)
    tw = _trapz_weights(w)
    lpivot2 = np.sum(tw * w * t) / np.sum(tw * t / w)
    return 10.0 ** (-0.4 * (2.5 * np.log10(lpivot2 / C_AA_PER_S) + 48.6))


def make_bands(tables, vega_wl, vega_flux):
    """Build a ``{name: Band}`` library from ``{name: (wavelength, transmit)}`` tables and a Vega
    spectrum: 2MASS bands get the Vega zero flux, everything else the AB zero flux (mft6.py:778-782)."""
    out = {}
    for name, (w, t) in tables.items():
        if '2MASS' in name:
            zero = Band(name, w, t, 1.0).mean_flux(vega_wl, vega_flux)
        else:
            zero = ab_zero_flux(w, t)
        out[name] = Band(name, w, t, zero)
    return out


def from_pyphot(lib, names=BAND_NAMES_6):
    """Adapt a pyphot library (``pyphot.get_library()``) to ``{name: Band}``."""
    out = {}
    for n in names:
        f = lib[n]
        w = np.asarray(getattr(f.wavelength, 'magnitude', f.wavelength), dtype=float)
        zero = f.Vega_zero_flux if '2MASS' in n else f.AB_zero_flux
        zero = float(getattr(zero, 'magnitude', getattr(zero, 'value', zero)))
        out[n] = Band(n, w, np.asarray(f.transmit, dtype=float), zero)
    return out
