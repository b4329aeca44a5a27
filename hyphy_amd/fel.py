"""FEL-style per-site fits on top of ``hyphy_hip_site_fits_evaluate``.

Host-side mirror of what ``fel.handle_a_site`` does for ONE site
(/root/reference/res/TemplateBatchFiles/SelectionAnalyses/FEL.bf:609-900): evaluate the starting grid
(FEL.bf:617-760), optimise the site's rate multipliers under the alternative model (alpha, beta_test,
beta_nuisance free), then under the null (beta_test := alpha), and form the likelihood-ratio test.  The
reference runs one single-site likelihood function per site through ``Optimize`` and farms sites out over
MPI (libv3/tasks/mpi.bf); here ALL sites advance in lockstep through a batched Nelder-Mead whose every
iteration is one device launch with four candidate parameter vectors per site (reflection, expansion and
the two contractions), so a whole alignment is fitted in a few hundred launches.

This module is host logic (numpy): the arithmetic of the likelihoods happens in the HIP library.
"""
from __future__ import annotations

import dataclasses
from typing import Optional, Sequence

import numpy as np

# FEL.bf:617-680 (site-rate-variation grid: alpha scaler included) as (alpha, beta) multiples
START_GRID = np.array([
    (0.01, 0.1), (1.0, 0.1), (1.0, 0.5), (1.0, 1.0), (1.0, 5.0), (10.0, 0.1),
    (0.01, 0.5), (0.01, 5.0), (10.0, 0.5), (10.0, 1.0), (10.0, 50.0), (100.0, 1.0),
])


@dataclasses.dataclass
class SiteFit:
    theta: np.ndarray        # [S, P] maximum-likelihood multipliers
    logl: np.ndarray         # [S] site log-likelihood at theta
    iterations: int
    launches: int
    converged: np.ndarray    # [S] bool


def _multipliers(theta: np.ndarray, param_map: np.ndarray) -> np.ndarray:
    """theta [..., S, P] -> site_mult [..., S, G, K]; param_map[g][k] = index into theta or -1 (fixed at 1)."""
    pm = np.asarray(param_map)
    out = np.ones(theta.shape[:-1] + pm.shape)
    for g in range(pm.shape[0]):
        for k in range(pm.shape[1]):
            if pm[g, k] >= 0:
                out[..., g, k] = theta[..., pm[g, k]]
    return out


def lockstep_nelder_mead(evaluate, S: int, P: int, start_u: np.ndarray, max_iter: int = 400, tol: float = 1e-9,
                         u_init: Optional[np.ndarray] = None, active: Optional[np.ndarray] = None):
    """Maximise f_s(u) for every site s at once: ``evaluate(U)`` maps candidate vectors U [n, S, P] (unconstrained
    coordinates) to values [n, S] — one device launch.  Every site starts from the best row of ``start_u`` [n_start, P]
    (or its row of ``u_init`` [S, P] when that is better).  Returns (u [S, P], f [S], iterations, converged [S])."""
    sp = np.asarray(start_u, dtype=np.float64)
    f0 = evaluate(np.broadcast_to(sp[:, None, :], (sp.shape[0], S, P)).copy())
    f0 = np.where(np.isnan(f0), -np.inf, f0)
    best = np.argmax(f0, axis=0)
    x0 = sp[best]                                                        # [S, P]
    f_best = f0[best, np.arange(S)]
    if u_init is not None:
        xi = np.asarray(u_init, dtype=np.float64)
        fi = evaluate(xi[None])[0]
        fi = np.where(np.isnan(fi), -np.inf, fi)
        use = fi > f_best
        x0 = np.where(use[:, None], xi, x0)
        f_best = np.where(use, fi, f_best)
    # initial simplex: x0 and x0 with one coordinate stretched
    X = np.repeat(x0[:, None, :], P + 1, axis=1)                         # [S, P+1, P]
    for k in range(P):
        X[:, k + 1, k] = X[:, k + 1, k] * 1.6 + 0.05
    F = np.empty((S, P + 1))
    F[:, 0] = f_best
    F[:, 1:] = evaluate(np.ascontiguousarray(np.transpose(X[:, 1:, :], (1, 0, 2)))).T
    F = np.where(np.isnan(F), -np.inf, F)
    rows = np.arange(S)
    it = 0
    converged = np.zeros(S, dtype=bool)
    for it in range(1, max_iter + 1):
        order = np.argsort(-F, axis=1)                                   # best first (maximisation)
        F = np.take_along_axis(F, order, axis=1)
        X = np.take_along_axis(X, order[:, :, None], axis=1)
        with np.errstate(invalid="ignore"):
            spread = F[:, 0] - F[:, -1]
        size = np.max(np.abs(X[:, 1:, :] - X[:, :1, :]), axis=(1, 2))
        converged = ((spread < tol) & (size < 1e-5)) | ~np.isfinite(F[:, 0])
        if active is not None:
            converged |= ~active
        if converged.all():
            break
        c = X[:, :-1, :].mean(axis=1)                                    # centroid of all but the worst
        w = X[:, -1, :]
        cand = np.stack([c + (c - w), c + 2.0 * (c - w), c + 0.5 * (c - w), c - 0.5 * (c - w)])   # r, e, oc, ic
        fc = evaluate(cand)
        fc = np.where(np.isnan(fc), -np.inf, fc)
        fr, fe, foc, fic = fc
        fb, fsw, fw = F[:, 0], F[:, -2], F[:, -1]
        newx = w.copy()
        newf = fw.copy()
        m_exp = fr > fb
        take_e = m_exp & (fe > fr)
        take_r = (m_exp & ~take_e) | (~m_exp & (fr > fsw))
        m_oc = ~m_exp & ~(fr > fsw) & (fr > fw)
        m_ic = ~m_exp & ~(fr > fsw) & ~(fr > fw)
        for mask, idx, fv in ((take_e, 1, fe), (take_r, 0, fr)):
            newx[mask] = cand[idx][mask]
            newf[mask] = fv[mask]
        ok_oc = m_oc & (foc >= fr)
        ok_ic = m_ic & (fic > fw)
        newx[ok_oc] = cand[2][ok_oc]
        newf[ok_oc] = foc[ok_oc]
        newx[ok_ic] = cand[3][ok_ic]
        newf[ok_ic] = fic[ok_ic]
        shrink = ((m_oc & ~ok_oc) | (m_ic & ~ok_ic)) & ~converged
        X[:, -1, :] = newx
        F[:, -1] = newf
        if shrink.any():
            Xs = X[:, :1, :] + 0.5 * (X - X[:, :1, :])
            fs = evaluate(np.ascontiguousarray(np.transpose(Xs[:, 1:, :], (1, 0, 2)))).T
            fs = np.where(np.isnan(fs), -np.inf, fs)
            X[shrink, 1:, :] = Xs[shrink, 1:, :]
            F[shrink, 1:] = fs[shrink]
    order = np.argmax(F, axis=1)
    return X[rows, order], F[rows, order], it, converged


def fit_sites(part, branch_group, branch_coeffs, root_freqs, param_map, start_points: np.ndarray,
              max_iter: int = 400, tol: float = 1e-9, upper: float = 1e3, x_init: Optional[np.ndarray] = None,
              active: Optional[np.ndarray] = None) -> SiteFit:
    """Maximise every site's log-likelihood over its own parameter vector theta (P entries, all >= 0).

    ``start_points`` [n_start, P]: every site starts from the best of these (one launch).  Lockstep Nelder-Mead in
    u = sqrt(theta) (keeps theta >= 0 without constraints, boundary optima theta = 0 are reachable); theta is capped at
    ``upper`` (the cost of an evaluation grows linearly with the rates, and the entry point refuses total rates > 4096).
    ``x_init`` [S, P]: an extra per-site starting point.  ``active`` [S] bool: only these sites are fitted — the others
    are evaluated with all-zero multipliers (the kernel skips zero-rate tiles) and come back with logl = nan."""
    pm = np.asarray(param_map, dtype=np.int64)
    S = part.S
    P = int(pm.max()) + 1
    launches = [0]

    def evaluate(U):  # U [n, S, P] -> logL [n, S]
        launches[0] += 1
        theta = np.minimum(U * U, upper)
        if active is not None:
            theta = np.where(active[None, :, None], theta, 0.0)
        return part.site_fits_evaluate(branch_group, branch_coeffs, _multipliers(theta, pm), root_freqs)

    u, f, it, converged = lockstep_nelder_mead(evaluate, S, P, np.sqrt(np.asarray(start_points, dtype=np.float64)),
                                               max_iter=max_iter, tol=tol,
                                               u_init=None if x_init is None else np.sqrt(np.asarray(x_init, dtype=np.float64)),
                                               active=active)
    if active is not None:
        f = np.where(active, f, np.nan)
    return SiteFit(theta=np.minimum(u * u, upper), logl=f, iterations=it, launches=launches[0], converged=converged)


@dataclasses.dataclass
class FelResult:
    alpha: np.ndarray
    beta: np.ndarray          # tested branches
    beta_nuisance: np.ndarray
    logl_alt: np.ndarray
    logl_null: np.ndarray
    lrt: np.ndarray
    p_value: np.ndarray
    launches: int


def fel(part, tested: Sequence[bool], syn_lengths, nonsyn_lengths, root_freqs, max_iter: int = 400,
        pattern_of_site: Optional[np.ndarray] = None) -> FelResult:
    """Per-site alpha / beta fits and the LRT for beta_test != alpha, for every pattern of ``part`` (whose templates
    must be (synonymous, non-synonymous), e.g. ``bench.templates_for(3)``).  ``tested[b]``: branch b belongs to the
    tested set; syn/nonsyn_lengths [B]: the branch's coefficients from the global fit (FEL.bf:560-590).  Returns
    per-pattern vectors, or per-site ones when ``pattern_of_site`` is given."""
    from scipy.stats import chi2
    tested = np.asarray(tested, dtype=bool)
    group = np.where(tested, 0, 1).astype(np.int64)
    bc = np.stack([np.asarray(syn_lengths, dtype=np.float64), np.asarray(nonsyn_lengths, dtype=np.float64)], axis=1)
    has_nuisance = bool((~tested).any())
    if has_nuisance:
        alt_map, null_map = np.array([[0, 1], [0, 2]]), np.array([[0, 0], [0, 1]])
        alt_start = np.array([(a, b, b) for a, b in START_GRID])
        null_start = np.array([(a, b) for a, b in START_GRID])
    else:
        alt_map, null_map = np.array([[0, 1], [0, 1]]), np.array([[0, 0], [0, 0]])
        alt_start = START_GRID.copy()
        null_start = np.unique(START_GRID[:, :1], axis=0)
    alt = fit_sites(part, group, bc, root_freqs, alt_map, alt_start, max_iter=max_iter)
    # the null is nested in the alternative: also start it from the alternative's optimum projected onto beta_test = alpha
    proj = np.stack([alt.theta[:, 0], alt.theta[:, 2]], axis=1) if has_nuisance else alt.theta[:, :1]
    null = fit_sites(part, group, bc, root_freqs, null_map, null_start, max_iter=max_iter)
    null_p = part.site_fits_evaluate(group, bc, _multipliers(proj, null_map), root_freqs)
    better = null_p > null.logl
    null.theta[better] = proj[better]
    null.logl[better] = null_p[better]
    launches = alt.launches + null.launches + 1
    # ... and the alternative contains the null: where the independent null fit ended above the alternative's (a
    # local optimum on a multi-modal surface), refit those sites' alternative from the embedded null optimum
    worse = null.logl > alt.logl + 1e-9
    if worse.any():
        emb = (np.stack([null.theta[:, 0], null.theta[:, 0], null.theta[:, 1]], axis=1) if has_nuisance
               else np.stack([null.theta[:, 0], null.theta[:, 0]], axis=1))
        again = fit_sites(part, group, bc, root_freqs, alt_map, alt_start[:1], max_iter=max_iter, x_init=emb, active=worse)
        launches += again.launches
        take = worse & (again.logl > alt.logl)
        alt.theta[take] = again.theta[take]
        alt.logl[take] = again.logl[take]
    lrt = np.maximum(0.0, 2.0 * (alt.logl - null.logl))
    res = FelResult(alpha=alt.theta[:, 0], beta=alt.theta[:, 1], beta_nuisance=alt.theta[:, 2] if has_nuisance else alt.theta[:, 1],
                    logl_alt=alt.logl, logl_null=null.logl, lrt=lrt, p_value=chi2.sf(lrt, 1),
                    launches=launches)
    if pattern_of_site is not None:
        idx = np.asarray(pattern_of_site)
        res = FelResult(**{f.name: (getattr(res, f.name)[idx] if f.name != "launches" else res.launches)
                           for f in dataclasses.fields(FelResult)})
    return res


# ---- MEME-style episodic selection: two omega classes per site, mixed on every tested branch -------------------------
@dataclasses.dataclass
class MemeResult:
    alpha: np.ndarray
    beta_minus: np.ndarray
    beta_plus: np.ndarray
    weight_minus: np.ndarray   # mixture weight of the beta_minus class
    beta_nuisance: np.ndarray  # untested branches
    logl_alt: np.ndarray
    logl_null: np.ndarray      # beta_plus constrained to <= alpha
    lrt: np.ndarray
    p_value: np.ndarray
    launches: int


def meme(part, tested: Sequence[bool], syn_lengths, nonsyn_lengths, root_freqs, max_iter: int = 400,
         upper: float = 1e3) -> MemeResult:
    """Per-site mixed-effects model of episodic selection in the manner of
    /root/reference/res/TemplateBatchFiles/SelectionAnalyses/MEME.bf: on the tested branches every site has two
    non-synonymous rate classes, beta_minus <= alpha with weight q and beta_plus (unconstrained) with weight 1 - q, mixed
    per BRANCH (P_b = q exp(Q_b^-) + (1 - q) exp(Q_b^+): the explicit-form route), untested branches evolve with
    (alpha, beta_nuisance) (MEME.bf:71-79, 487-508; the reference scales EVERY site rate by the branch's total length MLE, MEME.bf:837-845: pass that as both
    coefficient vectors to reproduce it).  Null: beta_plus <= alpha (the reference: beta_plus := alpha where the alternative has
    beta_plus > alpha, no test otherwise, MEME.bf:1432-1436).  p-value 2/3 - 2/3 (0.45 F_chi2_1 + 0.55 F_chi2_2) (MEME.bf:1656).  Lockstep Nelder-Mead over hyphy_hip_site_fits_evaluate_mixture."""
    from scipy.stats import chi2
    tested = np.asarray(tested, dtype=bool)
    group = np.where(tested, 0, 1).astype(np.int64)
    bc = np.stack([np.asarray(syn_lengths, dtype=np.float64), np.asarray(nonsyn_lengths, dtype=np.float64)], axis=1)
    S = part.S
    launches = [0]

    def unit(x):  # R -> [0, 1)
        return x * x / (1.0 + x * x)

    def make_eval(null):
        def evaluate(U):  # U [n, S, 5]: sqrt(alpha), omega_minus, beta_plus (or its ratio to alpha under the null), q, sqrt(beta_nuisance)
            launches[0] += 1
            alpha = np.minimum(U[..., 0] ** 2, upper)
            bminus = alpha * unit(U[..., 1])
            bplus = alpha * unit(U[..., 2]) if null else np.minimum(U[..., 2] ** 2, upper)
            q = unit(U[..., 3])
            sm = np.empty(U.shape[:2] + (2, 2, 2))            # [n, S, component, group, template]
            sm[..., 0] = alpha[..., None, None]
            sm[..., 0, 0, 1] = bminus                          # component "-", tested
            sm[..., 1, 0, 1] = bplus                           # component "+", tested
            sm[..., :, 1, 1] = np.minimum(U[..., 4] ** 2, upper)[..., None]   # untested branches: (alpha, beta_nuisance) in both components
            sw = np.stack([q, 1.0 - q], axis=-1)
            return part.site_fits_evaluate_mixture(group, bc, sm, sw, root_freqs)
        return evaluate

    def theta_of(u, null):
        alpha = np.minimum(u[:, 0] ** 2, upper)
        return (alpha, alpha * unit(u[:, 1]), (alpha * unit(u[:, 2]) if null else np.minimum(u[:, 2] ** 2, upper)), unit(u[:, 3]),
                np.minimum(u[:, 4] ** 2, upper))

    inv_unit = lambda y: np.sqrt(y / (1.0 - y))
    start_alt = np.array([(np.sqrt(a), inv_unit(w), np.sqrt(bp), inv_unit(q), np.sqrt(bn)) for a in (0.1, 1.0, 5.0)
                          for w in (0.1, 0.8) for bp in (0.5, 5.0) for q in (0.5, 0.9) for bn in (0.3, 3.0)])
    u_alt, f_alt, _, _ = lockstep_nelder_mead(make_eval(False), S, 5, start_alt, max_iter=max_iter)
    a_alt, bm_alt, bp_alt, q_alt, bn_alt = theta_of(u_alt, False)
    # null from the alternative's optimum projected onto beta_plus <= alpha, and from a small grid
    ratio = np.clip(bp_alt / np.maximum(a_alt, 1e-300), 0.0, 0.999)
    u_proj = np.stack([u_alt[:, 0], u_alt[:, 1], inv_unit(ratio), u_alt[:, 3], u_alt[:, 4]], axis=1)
    start_null = np.array([(np.sqrt(a), inv_unit(w), inv_unit(r), inv_unit(0.5), np.sqrt(bn)) for a in (0.1, 1.0, 5.0)
                           for w in (0.1, 0.8) for r in (0.5, 0.95) for bn in (0.3, 3.0)])
    u_null, f_null, _, _ = lockstep_nelder_mead(make_eval(True), S, 5, start_null, max_iter=max_iter, u_init=u_proj)
    # the alternative contains the null
    worse = f_null > f_alt + 1e-9
    if worse.any():
        a0, bm0, bp0, q0, bn0 = theta_of(u_null, True)
        emb = np.stack([u_null[:, 0], u_null[:, 1], np.sqrt(bp0), u_null[:, 3], u_null[:, 4]], axis=1)
        u2, f2, _, _ = lockstep_nelder_mead(make_eval(False), S, 5, start_alt[:1], max_iter=max_iter, u_init=emb)
        take = worse & (f2 > f_alt)
        u_alt[take], f_alt[take] = u2[take], f2[take]
        a_alt, bm_alt, bp_alt, q_alt, bn_alt = theta_of(u_alt, False)
    lrt = np.maximum(0.0, 2.0 * (f_alt - f_null))
    pv = (2.0 / 3.0) * (0.45 * chi2.sf(lrt, 1) + 0.55 * chi2.sf(lrt, 2))   # MEME.bf:1656 (2/3 at LRT = 0)
    return MemeResult(alpha=a_alt, beta_minus=bm_alt, beta_plus=bp_alt, weight_minus=q_alt, beta_nuisance=bn_alt, logl_alt=f_alt, logl_null=f_null,
                      lrt=lrt, p_value=pv, launches=launches[0])
