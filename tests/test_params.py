"""The parameter-file reader and data preparation (``mcmc_spec_amd.params``) against the NumPy statements of the
reference's ``main`` (mft6.py:3458-3466 parser, :3496-3499 telluric mask, :3502-3507 crop + normalise,
:3523-3541 list parsing, :3553-3562 photometry conversion).  The parameter text below is written by the test
in the reference's format (one space between key and value, value ends at the first tab, '#' comments,
arrays without spaces)."""
import os

import numpy as np
import pytest

from mcmc_spec_amd import params

PARTEXT = (
    "### a comment line\n"
    "\n"
    "models btsettl\t\t\t#which models\n"
    "dirname koiX\t\t\t#output directory\n"
    "fname X \t\t#value followed by a space then tabs\n"
    "res 1700 \t\t\t#resolution\n"
    "tmin 3000 \t#K\n"
    "tmax 4200 \t#K\n"
    "lgmin 4\t\t#ignored by main\n"
    "lgmax 5\t\t#ignored by main\n"
    "specmin 5000 \t#A\n"
    "specmax 24000\t#A\n"
    "mask True\n"
    "rad_prior tRUE\n"
    "filename spec.txt   #three spaces then a comment\n"
    "spmin 0.645 \t\t#um\n"
    "spmax 0.84\t\t#um\n"
    "cmag [2.08,1.3]\t\t#contrasts\n"
    "cerr [0.14,0.02]\t\t#errors\n"
    "cfilt ['lp600','Kp']\t\t#filters\n"
    "pmag [13.815,13.505,13.355,12.323,11.826,11.735] \t#mags\n"
    "perr [0.1,0.1,0.1,0.026,0.022,0.019]\t#errors\n"
    "pfilt ['sdss,r','sdss,i','sdss,z','j','h','k']\t\t#filters\n"
    "plx 2.0732e-3\t\t#arcsec\n"
    "plx_err 0.0277e-3 \t\t#arcsec\n"
    "dist_fit True\n"
    "av 0.106\t\t#mag\n"
    "av_err 0.01 \t\t#mag\n"
    "ra 288.456118\n"
    "dec 45.802226\n"
    "nwalk 150 \t\t#walkers\n"
    "nstep 400\t\t#steps\n"
    "nspec 2\t\t#stars\n"
    "ndust 0\t\t#disks\n"
    "nburn 301\t\t#burn-in\n"
    "nsteps 15001\t\t#steps\n"
    "real_values [0,0,0,0,0,0,0,0]\n")


@pytest.fixture()
def run_dir(tmp_path):
    rng = np.random.default_rng(4)
    wl = np.concatenate([np.linspace(0.60, 0.70, 400), np.linspace(0.695, 0.90, 700)])  # two overlapping arms
    flux = 1e-15 * (1 + 0.1 * rng.normal(size=wl.size)) * (wl / 0.7) ** 2
    err = 1e-17 * (1 + rng.uniform(size=wl.size))
    np.savetxt(tmp_path / 'spec.txt', np.column_stack((wl, flux, err)))
    (tmp_path / 'param_test.txt').write_text(PARTEXT)
    return tmp_path, wl, flux, err


def test_parser_quirks(run_dir):
    d, *_ = run_dir
    pd = params.read_param_file(d / 'param_test.txt')
    assert pd['models'] == 'btsettl' and pd['fname'] == 'X' and pd['res'] == '1700'
    assert pd['lgmin'] == '4' and pd['specmax'] == '24000'
    assert pd['mask'] == 'True\n'                      # no tab on the line: the newline stays (mft6.py:3463-3464)
    assert pd['filename'] == 'spec.txt'                # split(' ')[0:2] drops everything after the second space
    assert pd['pfilt'] == "['sdss,r','sdss,i','sdss,z','j','h','k']"
    assert '###' not in pd and len(pd) == 35
    with pytest.raises(ValueError):                    # a key with no space cannot be unpacked, as in the reference
        (d / 'bad.txt').write_text('keyonly\n')
        params.read_param_file(d / 'bad.txt')


def test_boolean_and_list_conventions():
    assert params.truthy('True\n') and params.truthy('t') and params.truthy('tRUE') and params.truthy('yes it is')
    assert not params.truthy('False\n') and not params.truthy('f') and not params.truthy('0')
    assert params.float_list('[2.08,1.3]') == [2.08, 1.3] and params.float_list('[]') == [] and list(params.name_list('[]')) == []
    v = params.float_list('[13.8,np.nan,12.3]')       # mft6.py:3530-3536
    assert v[0] == 13.8 and np.isnan(v[1]) and v[2] == 12.3
    with pytest.raises(ValueError):
        params.float_list('[1,abc]')
    assert list(params.name_list("['lp600','Kp']")) == ['lp600', 'Kp']
    assert list(params.name_list("['sdss,r','sdss,i','j']")) == ['sdss,r', 'sdss,i', 'j']


def test_mask_crop_and_normalisation_follow_the_reference_statements(run_dir):
    d, wl, flux, err = run_dir
    data_wl, dsp, de = np.genfromtxt(d / 'spec.txt', unpack=True)
    # mft6.py:3497-3499 (flux and error first, with the unmasked wavelengths; then the wavelengths)
    sel = lambda a: np.concatenate((a[np.where(data_wl <= 0.6860)], a[np.where((data_wl >= 0.6880) & (data_wl <= 0.7600))],
                                    a[np.where((data_wl >= 0.7660) & (data_wl <= 0.8210))], a[np.where(data_wl > 0.8240)]))
    dsp, de, data_wl = sel(dsp), sel(de), sel(data_wl)
    # mft6.py:3502-3507
    k = np.where((data_wl > 0.645) & (data_wl < 0.84))
    data_wl, dsp, de = data_wl[k], dsp[k], de[k]
    de /= np.median(dsp)
    dsp /= np.median(dsp)
    w, f, e = params.prepare_data(d / 'spec.txt', 0.645, 0.84, mask=True)
    assert np.array_equal(w, data_wl) and np.array_equal(f, dsp) and np.array_equal(e, de)
    assert not np.any((w > 0.6860) & (w < 0.6880)) and not np.any((w > 0.7600) & (w < 0.7660))
    assert np.sum(np.diff(w) <= 0) == 1               # the arm overlap survives: query wavelengths are not sorted
    assert abs(np.median(f) - 1.0) < 1e-15
    # exclusive crop: a pixel exactly at spmin / spmax is dropped
    w2, _, _ = params.prepare_data(d / 'spec.txt', wl[10], wl[-10], mask=False)
    assert wl[10] not in w2 and wl[-10] not in w2 and wl[11] in w2


def test_photometry_conversion_and_run_assembly(run_dir):
    d, *_ = run_dir
    run = params.load_run(str(d / 'param_test.txt'), data_root=str(d))
    assert run.mask and run.rad_prior and run.dist_fit and run.nspec == 2 and run.res == 1700
    assert run.logg_range == [4, 5.5] and run.specrange == [5000, 24000]
    old = [13.815, 13.505, 13.355, 12.323, 11.826, 11.735]
    # mft6.py:3556-3557 by hand: r uses r-i, i uses r-i, z uses i-z
    want = [-0.0383 + 0.0548 * (old[0] - old[1]) + old[0], -0.0583 + 0.0696 * (old[0] - old[1]) + old[1],
            -0.0597 + 0.1587 * (old[1] - old[2]) + old[2], old[3], old[4], old[5]]
    assert np.array_equal(run.fr[3], np.array(want))
    assert run.fr[0] == [2.08, 1.3] and list(run.fr[2]) == ['lp600', 'Kp'] and len(run.fr[5]) == 6
    assert run.prior(6) == [*np.zeros(10), 2.0732e-3, 0.0277e-3]   # mft6.py:3689
    assert run.r == [min(run.data[0]), max(run.data[0])]
    # a parameter file whose name contains 'synth' passes the magnitudes through (mft6.py:3553, 3562)
    os.rename(d / 'param_test.txt', d / 'param_synth.txt')
    run2 = params.load_run(str(d / 'param_synth.txt'), data_root=str(d))
    assert np.array_equal(run2.fr[3], np.array(old))
