"""How tightly a number the survey measured on the reference may be asserted.

tests/golden/reference_recorded.json holds per-history means the survey printed from ONE
reference run of n_ref histories.  Such a figure carries (i) the Monte-Carlo error of an
n_ref-history mean and (ii) the rounding of the printed value.  Our own mean comes from many
more histories, so the tolerance is

    3 sigma(n_ref)  +  half a unit of the last printed digit  (+ 3 sigma of our own mean)

with sigma(n_ref) estimated from OUR batch means at the reference's sample size: batches of
n_ref histories each, sigma = their standard deviation."""
import numpy as np


def half_unit_of_last_digit(value):
    """0.5e-k for a value printed with k decimals (173 -> 0.5, 27.9 -> 0.05, 0.04 -> 0.005)."""
    s = repr(float(value)) if not isinstance(value, str) else value
    if float(value) == int(float(value)) and "." in s and s.endswith(".0"):
        return 0.5
    decimals = len(s.split(".")[1]) if "." in s else 0
    return 0.5 * 10.0 ** (-decimals)


def tolerance(batch_means, printed_value, n_sigma=3.0):
    """Allowed |ours - printed| where `batch_means` are our per-history means over batches of the
    reference's sample size."""
    b = np.asarray(batch_means, dtype=float)
    sigma_ref = b.std(ddof=1)
    sigma_ours = sigma_ref / np.sqrt(len(b))
    return n_sigma * np.hypot(sigma_ref, sigma_ours) + half_unit_of_last_digit(printed_value)


def poisson_fraction_tolerance(frac_ref, n_ref, frac_ours, n_ours, n_sigma=3.0):
    """Allowed |ours - ref| between two event fractions counted in n_ref resp. n_ours trials."""
    return n_sigma * np.sqrt(frac_ref / n_ref + frac_ours / n_ours)
