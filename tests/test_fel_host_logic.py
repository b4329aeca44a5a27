"""Host logic of the FEL-style driver (hyphy_amd/fel.py) without a GPU: the lockstep Nelder-Mead and the
alternative / null bookkeeping, run against a small numpy stand-in for hyphy_hip_site_fits_evaluate (same
argument meaning: site_mult [sets, S, G, K], log-likelihood of every pattern under its own multipliers)."""
import numpy as np
import scipy.linalg
import scipy.optimize

from hyphy_amd import fel, tree


class _NumpySiteFits:
    """Stand-in with the interface of HipPartition.site_fits_evaluate (tests only; tiny 4-state trees)."""

    def __init__(self, flat, codes, T):
        self.flat, self.codes, self.T = flat, codes, T
        self.S, self.B, self.D = codes.shape[1], flat.n_branches, T.shape[1]

    def site_fits_evaluate(self, branch_group, branch_coeffs, site_mult, root_freqs):
        sm = np.asarray(site_mult, dtype=np.float64)
        single = sm.ndim == 3
        if single:
            sm = sm[None]
        out = np.zeros(sm.shape[:2])
        L, I, D = self.flat.L, self.flat.I, self.D
        idx = np.arange(D)
        for st in range(sm.shape[0]):
            for s in range(self.S):
                x = sm[st, s][np.asarray(branch_group)] * np.asarray(branch_coeffs)
                Q = np.einsum("bk,kij->bij", x, self.T)
                Q[:, idx, idx] = 0.0
                Q[:, idx, idx] = -Q.sum(2)
                P = np.stack([scipy.linalg.expm(q) for q in Q])
                cond = np.ones((I, D))
                for node in range(L + I - 1):
                    par = int(self.flat.flat_parents[node])
                    v = np.eye(D)[self.codes[node, s]] if node < L else cond[node - L]
                    cond[par] *= P[node] @ v
                out[st, s] = np.log(cond[I - 1] @ np.asarray(root_freqs))
        return out[0] if single else out


    def site_fits_evaluate_mixture(self, branch_group, branch_coeffs, site_mult, site_weights, root_freqs):
        sm, sw = np.asarray(site_mult, dtype=np.float64), np.asarray(site_weights, dtype=np.float64)
        out = np.zeros(sm.shape[:2])
        L, I, D = self.flat.L, self.flat.I, self.D
        idx = np.arange(D)
        for st in range(sm.shape[0]):
            for s in range(self.S):
                P = 0.0
                for m in range(sm.shape[2]):
                    Q = np.einsum("bk,kij->bij", sm[st, s, m][np.asarray(branch_group)] * np.asarray(branch_coeffs), self.T)
                    Q[:, idx, idx] = 0.0
                    Q[:, idx, idx] = -Q.sum(2)
                    P = P + sw[st, s, m] * np.stack([scipy.linalg.expm(q) for q in Q])
                cond = np.ones((I, D))
                for node in range(L + I - 1):
                    par = int(self.flat.flat_parents[node])
                    v = np.eye(D)[self.codes[node, s]] if node < L else cond[node - L]
                    cond[par] *= P[node] @ v
                out[st, s] = np.log(cond[I - 1] @ np.asarray(root_freqs))
        return out


def _toy(seed=0, taxa=5, sites=10):
    rng = np.random.default_rng(seed)
    flat = tree.flatten(tree.random_tree(taxa, rng))
    D = 4
    pi = np.array([0.3, 0.2, 0.25, 0.25])
    T = np.zeros((2, D, D))
    T[0, 0, 2] = T[0, 2, 0] = T[0, 1, 3] = T[0, 3, 1] = 1.0           # "synonymous": transitions
    T[1] = 1.0 - np.eye(D) - T[0]                                      # "non-synonymous": transversions
    T = T * pi[None, None, :]
    base = rng.integers(0, D, size=sites)
    codes = np.where(rng.random((flat.L, sites)) < 0.2, rng.integers(0, D, size=(flat.L, sites)), base[None, :])
    codes[:, 0] = base[0]                                              # an invariable site: both rates -> 0
    return flat, codes, T, pi, rng


def test_lockstep_nelder_mead_reaches_the_per_site_optima():
    flat, codes, T, pi, rng = _toy()
    part = _NumpySiteFits(flat, codes, T)
    B = flat.n_branches
    group = np.zeros(B, dtype=np.int64)
    bc = np.stack([np.full(B, 0.2), np.full(B, 0.1)], axis=1)
    fit = fel.fit_sites(part, group, bc, pi, np.array([[0, 1]]), fel.START_GRID, max_iter=200)
    assert fit.theta.shape == (part.S, 2) and (fit.theta >= 0).all()
    for s in range(part.S):
        def neg(u, s=s):
            sm = np.ones((part.S, 1, 2))
            sm[s, 0] = u * u
            return -part.site_fits_evaluate(group, bc, sm, pi)[s]
        best = min((scipy.optimize.minimize(neg, np.sqrt(x0), method="Nelder-Mead",
                                            options=dict(xatol=1e-7, fatol=1e-10, maxiter=600)) for x0 in fel.START_GRID[[3, 7]]),
                   key=lambda r: r.fun)
        assert fit.logl[s] >= -best.fun - 1e-6, (s, fit.logl[s], -best.fun)
    # the invariable site: every substitution only lowers the likelihood
    assert fit.theta[0].max() < 1e-3


def test_fel_alternative_contains_the_null():
    flat, codes, T, pi, rng = _toy(seed=3, taxa=6, sites=8)
    part = _NumpySiteFits(flat, codes, T)
    B = flat.n_branches
    tested = rng.random(B) < 0.5
    tested[0], tested[1] = True, False
    res = fel.fel(part, tested, np.full(B, 0.15), np.full(B, 0.1), pi, max_iter=250, pattern_of_site=np.arange(part.S)[::-1])
    assert res.alpha.shape == (part.S,)
    assert (res.logl_alt >= res.logl_null - 1e-7).all()
    assert ((res.p_value >= 0) & (res.p_value <= 1)).all()
    assert np.allclose(res.lrt, np.maximum(0, 2 * (res.logl_alt - res.logl_null)))


def test_meme_driver_bookkeeping():
    flat, codes, T, pi, rng = _toy(seed=5, taxa=5, sites=4)
    part = _NumpySiteFits(flat, codes, T)
    B = flat.n_branches
    tested = np.ones(B, dtype=bool)
    tested[-1] = False
    res = fel.meme(part, tested, np.full(B, 0.2), np.full(B, 0.1), pi, max_iter=60)
    assert res.alpha.shape == (part.S,)
    assert (res.beta_minus <= res.alpha + 1e-12).all() and ((res.weight_minus >= 0) & (res.weight_minus <= 1)).all()
    assert (res.logl_alt >= res.logl_null - 1e-7).all()
    assert ((res.p_value >= 0) & (res.p_value <= 1)).all()
    # the reported optimum reproduces through an independent evaluation of the mixture entry point
    group = np.where(tested, 0, 1)
    sm = np.empty((part.S, 2, 2, 2))
    sm[..., 0] = res.alpha[:, None, None]
    sm[:, 0, 0, 1], sm[:, 1, 0, 1] = res.beta_minus, res.beta_plus
    sm[:, :, 1, 1] = res.beta_nuisance[:, None]
    sw = np.stack([res.weight_minus, 1 - res.weight_minus], axis=1)
    bc = np.stack([np.full(B, 0.2), np.full(B, 0.1)], axis=1)
    again = part.site_fits_evaluate_mixture(group, bc, sm[None], sw[None], pi)[0]
    assert np.allclose(again, res.logl_alt, rtol=0, atol=1e-9)


def test_bench_reads_the_measured_instruction_peak_from_the_committed_microbenchmark():
    """bench.py's `instruction_peak_measured` is read from a committed microbenchmark (no literals in the line) — since
    r04 the VGPR-accumulator run, profiles/r04_ubench_mfma_agpr_vs_vgpr.txt (r01's file measured the AGPR form: half the rate)."""
    import bench
    peak, src = bench.measured_instruction_peak()
    assert src == "profiles/r04_ubench_mfma_agpr_vs_vgpr.txt" and 70.0 < peak <= 78.6
    flops, bytes_ = bench.alg_work(61, 9974, 64, 62)
    assert flops == 4642198820 and bytes_ == 599317712      # SURVEY 8d at the headline size


def test_fubar_fixture_through_the_site_fit_interface():
    """tests/golden/ref_fubar_12x60.npz (site log-likelihoods of the reference's own FUBAR.bf on its rate grid) through the
    argument conventions of hyphy_hip_site_fits_evaluate — here with the numpy stand-in on a few grid points and sites (scipy's
    expm, 61 states); the GPU test runs the same helper over the whole 100 x 60 matrix on the device."""
    from tests import common
    fx = common.load("ref_fubar_12x60")
    gp, st = [0, 11, 37, 99], [0, 1, 7, 30]
    T, group, coeffs, mult, codes, want = common.fubar_site_fit_args(fx, gp, st)
    flat = tree.flat_from_parents(fx["flat_parents"], int(fx["L"]))
    part = _NumpySiteFits(flat, codes, T)
    got = part.site_fits_evaluate(group, coeffs, mult, fx["root_freqs"])
    fin = np.isfinite(want)
    assert fin.sum() >= 12
    assert np.max(np.abs(got[fin] - want[fin]) / np.abs(want[fin])) < 1e-9
