"""Synthetic workloads of the measurement plan (SURVEY.md 8d / BASELINE.json configs): the
shipped parameter box sampled with seed 42, and the Beer-Lambert excitation profiles that the
reference's example excitation files contain (`Example Data/Power_scan_Excitations.csv`,
`Twothick_Excitations.csv`: dN(x) = A exp(-alpha x), alpha = 6.0e-3 / nm, x = (n + 1/2) dx),
regenerated analytically so that any grid size L can be produced without the data files."""
import numpy as np

from .sampler import default_box

ALPHA_PER_NM = 6.0e-3
# peak excess densities of the three Power_scan excitations, cm^-3 (fit of the shipped rows,
# residual 4e-9)
POWER_SCAN_A_CM3 = (1.2738364686918604e16, 1.1539459772368606e17, 1.648494253696512e18)
CM3_TO_NM3 = 1e-21                       # bayes_io.get_initpoints scale_f (bayes_io.py:106)


def beer_lambert(A_cm3, length_nm, L, alpha=ALPHA_PER_NM):
    x = (np.arange(L) + 0.5) * (length_nm / L)
    return A_cm3 * CM3_TO_NM3 * np.exp(-alpha * x)


def power_scan(L=128, length_nm=2000.0):
    """3 excitations, one thickness (parallel_bayes_gpu.py:72 with Length = 2000).
    Returns (init_params (3,L) nm^-3, lengths (3,))."""
    ini = np.stack([beer_lambert(A, length_nm, L) for A in POWER_SCAN_A_CM3])
    return ini, np.full(3, float(length_nm))


def twothick(L=128, lengths_nm=(311.0, 2000.0)):
    """6 curves: the three powers at alternating thickness 311 / 2000 nm
    (parallel_bayes_gpu.py:71)."""
    ini, lens = [], []
    for A in POWER_SCAN_A_CM3:
        for ln in lengths_nm:
            ini.append(beer_lambert(A, ln, L))
            lens.append(float(ln))
    return np.stack(ini), np.array(lens)


# marked point of the reference's Visualization/config.txt:57-68 (common units), used to
# synthesise observations because the shipped observation files for these two scans are absent
MARKED_POINT = np.array([1e8, 3e15, 20, 20, 4.8e-11, 2, 2, 4.4e-29, 4.4e-29, 511, 871, 0.1, 0])


def samples(S, seed=42):
    """(S,13) parameter sample in solver units, the reference's draw order and seed."""
    return default_box(seed, S)
