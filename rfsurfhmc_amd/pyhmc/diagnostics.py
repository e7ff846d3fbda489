"""Convergence diagnostics across the (many) independent chains of a batched run (the reference has
none -- its chains are separate MPI ranks that only meet in misfit.npy).

    samples: [nchains, nsamples, nparam]   e.g. sampler.x_cache, or the "model" array of a {name}.rank{r}.npz

* split_rhat   rank-free split-R-hat of Gelman et al. (BDA3 11.4): every chain is cut in halves, R-hat compares
               the between-half-chain variance with the within variance, per parameter.
* ess          effective sample size per parameter from the chain-averaged autocorrelation (Geyer's initial positive
               sequence), autocovariances by FFT, all chains at once.
* summarize    mean, sd, quantiles, R-hat, ESS per parameter in one dict of arrays.
Pure numpy, vectorised over chains and parameters; 8192 chains x 800 samples x 60 parameters take seconds."""
import numpy as np


def _split(samples):
    x = np.asarray(samples, dtype=np.float64)
    if x.ndim == 2:
        x = x[:, :, None]
    nc, ns, npar = x.shape
    h = ns // 2
    if h < 2:
        raise ValueError("need at least 4 samples per chain")
    return np.concatenate((x[:, :h], x[:, ns - h:]), axis=0)        # [2 nc, h, npar]


def split_rhat(samples):
    """Split-R-hat per parameter; values close to 1 (commonly < 1.01) indicate that the chains agree."""
    x = _split(samples)
    m, n, _ = x.shape
    cm = x.mean(axis=1)                                             # [m, npar]
    W = x.var(axis=1, ddof=1).mean(axis=0)
    B = n * cm.var(axis=0, ddof=1)
    var_plus = (n - 1) / n * W + B / n
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.sqrt(var_plus / W)


def _autocov(x):
    """Autocovariance along axis 1 of [m, n, npar] by FFT (biased estimator, as usual for ESS)."""
    m, n, _ = x.shape
    nfft = 1 << (2 * n - 1).bit_length()
    xc = x - x.mean(axis=1, keepdims=True)
    f = np.fft.rfft(xc, nfft, axis=1)
    ac = np.fft.irfft(f * np.conj(f), nfft, axis=1)[:, :n]
    return ac / n


def ess(samples):
    """Effective sample size per parameter over all chains (split chains, Geyer truncation)."""
    x = _split(samples)
    m, n, npar = x.shape
    ac = _autocov(x)                                                # [m, n, npar]
    W = x.var(axis=1, ddof=1).mean(axis=0)
    cm = x.mean(axis=1)
    B_over_n = cm.var(axis=0, ddof=1) if m > 1 else np.zeros(npar)
    var_plus = (n - 1) / n * W + B_over_n
    with np.errstate(divide="ignore", invalid="ignore"):
        rho = 1.0 - (W - ac.mean(axis=0)) / var_plus                # [n, npar], rho[0] = 1 up to the variance estimators
    rho[0] = 1.0
    # Geyer: sum consecutive pairs while they stay positive
    npair = n // 2
    pairs = rho[0:2 * npair:2] + rho[1:2 * npair:2]                 # [npair, npar]
    pos = np.cumprod(pairs > 0, axis=0).astype(bool)
    pairs = np.where(pos, pairs, 0.0)
    pairs = np.minimum.accumulate(pairs, axis=0)                    # initial monotone sequence
    tau = -1.0 + 2.0 * pairs.sum(axis=0)
    tau = np.maximum(tau, 1.0 / np.log10(max(m * n, 10)))
    return m * n / tau


def summarize(samples, quantiles=(0.025, 0.5, 0.975)):
    x = np.asarray(samples, dtype=np.float64)
    if x.ndim == 2:
        x = x[:, :, None]
    flat = x.reshape(-1, x.shape[2])
    out = {"mean": flat.mean(axis=0), "sd": flat.std(axis=0, ddof=1), "rhat": split_rhat(x), "ess": ess(x)}
    for q in quantiles:
        out[f"q{q:g}"] = np.quantile(flat, q, axis=0)
    return out
