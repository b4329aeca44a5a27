"""TEST INFRASTRUCTURE — generates the golden fixtures under ``tests/golden/`` by running the
REAL reference binary (``oracle/_ref/hyphy``; build it with ``make -f oracle/Makefile.ref``)
on deterministic synthetic inputs.  Run in the build container only (needs
/root/reference for the build); the fixtures it writes are plain data (inputs + expected
outputs) and travel with the repo.

    python -m oracle.make_golden            # regenerate everything

Cases (SURVEY §8c "Fixtures to commit"):
  codon_small   8 taxa x 40 codons, MG94xREV, logL + per-site logL
  codon_ambig   same tree, 5% gaps / N / partial ambiguities
  codon_deep    120-taxon ladder tree, longer branches -> forces 2^64 rescaling
  codon_wide    random 64-taxon tree (the benchmark's shape) x 60 codons
  nuc_wide      random 100-taxon tree x 200 sites
  codon_cat3    3 discrete rate classes (weights .7/.25/.05, rates .1/1/5) -> category mixing
  nuc_small     HKY85, 8 taxa x 300 sites (pattern compression, freq > 1)
  nuc_ambig     GTR with ambiguities
  nuc_deep      300-taxon ladder, rescaling in the 4-state path
  expm_*        P = Exp(Q) through _Matrix::Exponentiate for 4/20/61-state Q at several scales
  codon_mix2/3  branch-site mixtures in the reference's explicit form (sum_m w_m Exp(Q_m) on every branch): BUSTED / BS-REL shape
  ref_smallcodon  the reference's own known-answer test SimpleOptimizations/SmallCodon.bf (data + expected log L)
  ref_fluHA       real data of SimpleOptimizations/IntermediateNuc.bf (HKY85, 349 influenza sequences: the 4-state path)
  ref_busted_16x150 (python -m oracle.make_golden busted) the reference's unmodified BUSTED.bf on a simulated 16 x 150 codon
                  alignment: MLEs of the unconstrained branch-site mixture + the log L at them (explicit-form path)
  ref_fubar_12x60 (python -m oracle.make_golden fubar) the reference's unmodified FUBAR.bf: site log-likelihoods of a 12 x 60 codon
                  alignment on its 10 x 10 (alpha, beta) grid, recovered from FUBAR's cache file (per-site batched evaluation)
  ref_fel_12x60   (python -m oracle.make_golden fel) the reference's unmodified FEL.bf on a 12 x 60 codon alignment: per-site
                  alpha / beta / LRT / p-value table + the global fit its site phase starts from
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hyphy_amd import data, models, tree  # noqa: E402
from oracle import hbl  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

POS_FREQS = np.array([[0.30, 0.20, 0.25, 0.25], [0.20, 0.30, 0.30, 0.20], [0.25, 0.25, 0.20, 0.30]])
REV = dict(AC=0.5, AT=0.4, CG=0.4, CT=1.2, GT=0.4)
NUC_FREQS = np.array([0.35, 0.15, 0.2, 0.3])


def branch_lengths(flat, seed, lo, hi):
    rng = np.random.default_rng(seed)
    return {n: float(rng.uniform(lo, hi)) for n in flat.branch_names()}


def codon_case(name, n_taxa, n_codons, seed, *, ladder=False, missing=0.0, tlo=0.02, thi=0.12,
               omega=0.3, category=None, p_change=0.04):
    tr = tree.caterpillar_tree(n_taxa) if ladder else None
    syn = data.evolve(n_taxa, n_codons, 3, seed=seed, p_change=p_change, tree=tr)
    seqs = syn.seqs
    if missing > 0:
        seqs = data.inject_missing(seqs, 3, missing, seed + 1000)
    flat = syn.flat
    bt = branch_lengths(flat, seed + 7, tlo, thi)
    tmpl = models.mg94rev_template(POS_FREQS)
    pi = models.f3x4_codon_freqs(POS_FREQS)
    g = dict(R=omega, **REV)
    rate_expr = "t" if category is None else f"{category['name']}*t"
    res = hbl.evaluate(names=flat.leaf_names, seqs=seqs, newick=tree.to_newick(syn.tree), unit=3,
                       model_block=hbl.codon_model_block(tmpl, pi, rate_expr=rate_expr), model_name="MGM",
                       globals_=g, branch_t=bt, category=category)
    pd = data.compress(seqs, 3)
    fx = dict(kind="codon", D=61, L=flat.L, flat_parents=flat.flat_parents, leaf_codes=pd.leaf_codes,
              ambig=pd.ambig, pattern_freq=pd.pattern_freq, site_to_pattern=pd.site_to_pattern,
              t=np.array([bt[n] for n in flat.branch_names()]), omega=omega,
              rev=np.array([REV[k] for k in ("AC", "AT", "CG", "CT", "GT")]), pos_freqs=POS_FREQS,
              root_freqs=pi, logl=res["logl"], site_logl=res["site_logl"])
    if category:
        fx["cat_weights"] = np.array(category["weights"])
        fx["cat_values"] = np.array(category["values"])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print(f"{name}: logL = {res['logl']!r}  S = {pd.S}  n_ambig = {len(pd.ambig)}")


def nuc_case(name, n_taxa, n_sites, seed, *, ladder=False, missing=0.0, rev=None, tlo=0.02, thi=0.2, p_change=0.04):
    tr = tree.caterpillar_tree(n_taxa) if ladder else None
    syn = data.evolve(n_taxa, n_sites, 1, seed=seed, tree=tr, p_change=p_change)
    seqs = syn.seqs
    if missing > 0:
        seqs = data.inject_missing(seqs, 1, missing, seed + 1000)
    flat = syn.flat
    bt = branch_lengths(flat, seed + 7, tlo, thi)
    rev = rev or models.hky85_rev(0.35)
    g = {k: v for k, v in rev.items()}
    res = hbl.evaluate(names=flat.leaf_names, seqs=seqs, newick=tree.to_newick(syn.tree), unit=1,
                       model_block=hbl.nuc_model_block(NUC_FREQS), model_name="NM", globals_=g, branch_t=bt)
    pd = data.compress(seqs, 1)
    fx = dict(kind="nuc", D=4, L=flat.L, flat_parents=flat.flat_parents, leaf_codes=pd.leaf_codes,
              ambig=pd.ambig, pattern_freq=pd.pattern_freq, site_to_pattern=pd.site_to_pattern,
              t=np.array([bt[n] for n in flat.branch_names()]),
              rev=np.array([rev[k] for k in ("AC", "AT", "CG", "CT", "GT")]),
              root_freqs=NUC_FREQS, logl=res["logl"], site_logl=res["site_logl"])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print(f"{name}: logL = {res['logl']!r}  S = {pd.S}  n_ambig = {len(pd.ambig)}")


def mixture_case(name="codon_mix2", n_taxa=12, n_codons=60, seed=41, omegas=(0.1, 2.5), weights=(0.7, 0.3)):
    """Branch-site mixture in the reference's explicit form (BUSTED / BS-REL shape): every branch's transition matrix is
    sum_m w_m Exp(Q_m), Q_m = MG94xREV with omega_m (`hbl.codon_mixture_model_block`)."""
    syn = data.evolve(n_taxa, n_codons, 3, seed=seed)
    flat = syn.flat
    bt = branch_lengths(flat, seed + 7, 0.02, 0.3)
    tmpl = models.mg94rev_template(POS_FREQS)
    pi = models.f3x4_codon_freqs(POS_FREQS)
    M = len(omegas)
    g = dict(REV)
    for m, om in enumerate(omegas, start=1):
        g[f"R{m}"] = om
    for m, w in enumerate(weights[:-1], start=1):
        g[f"W{m}"] = w
    wexpr = [f"W{m}" for m in range(1, M)] + ["(1" + "".join(f"-W{m}" for m in range(1, M)) + ")"]
    block = hbl.codon_mixture_model_block(tmpl, pi, [f"R{m}" for m in range(1, M + 1)], wexpr)
    res = hbl.evaluate(names=flat.leaf_names, seqs=syn.seqs, newick=tree.to_newick(syn.tree), unit=3, model_block=block,
                       model_name="MGM", globals_=g, branch_t=bt)
    pd = data.compress(syn.seqs, 3)
    fx = dict(kind="codon_mixture", D=61, L=flat.L, flat_parents=flat.flat_parents, leaf_codes=pd.leaf_codes,
              ambig=pd.ambig, pattern_freq=pd.pattern_freq, site_to_pattern=pd.site_to_pattern,
              t=np.array([bt[n] for n in flat.branch_names()]), omegas=np.array(omegas), weights=np.array(weights),
              rev=np.array([REV[k] for k in ("AC", "AT", "CG", "CT", "GT")]), pos_freqs=POS_FREQS,
              root_freqs=pi, logl=res["logl"], site_logl=res["site_logl"])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print(f"{name}: logL = {res['logl']!r}  S = {pd.S}")


def expm_cases():
    rng = np.random.default_rng(5)
    Qs, Ps, names = [], [], []
    for t in (0.001, 0.05, 0.7, 6.0):
        Q = models.mg94rev_Q(t, 0.45, REV, POS_FREQS)
        Qs.append(Q); names.append(f"mg94_t{t}")
    for t in (0.01, 0.5, 20.0):
        Q = models.nuc_rev_Q(t, models.hky85_rev(0.3), NUC_FREQS)
        Qs.append(Q); names.append(f"hky_t{t}")
    for t in (0.02, 1.5):   # 20-state random reversible
        pi = rng.dirichlet(np.ones(20) * 5)
        Sx = rng.uniform(0.1, 2.0, size=(20, 20)); Sx = (Sx + Sx.T) / 2
        Q = Sx * pi[None, :] * t
        np.fill_diagonal(Q, 0.0)
        Q = models.finish_rate_matrix(Q)
        Qs.append(Q); names.append(f"aa_t{t}")
    out = {}
    for n, Q in zip(names, Qs):
        P = hbl.expm_via_reference(Q)
        out["Q_" + n] = Q
        out["P_" + n] = P
        print(f"expm {n}: row-sum err {abs(P.sum(1) - 1).max():.2e}")
    np.savez_compressed(os.path.join(OUT, "expm.npz"), **out)


def marginal_support_case(name="codon_small_marginal", n_taxa=8, n_codons=40, seed=11):
    """Pinned-state evaluations (ComputeBlock with branchIndex >= 0): the support matrix that
    ReconstructAncestors (lf, MARGINAL) leaves in <dataset>.marginal_support_matrix
    (likefunc2.cpp:932-1040): rows = internal nodes (in-order index), columns = pattern * D + state,
    value = L_s(node pinned to state) / L_s; the last state of every pattern is left at 0 by the reference."""
    import tempfile
    syn = data.evolve(n_taxa, n_codons, 3, seed=seed)
    flat = syn.flat
    bt = branch_lengths(flat, seed + 7, 0.02, 0.12)
    tmpl = models.mg94rev_template(POS_FREQS)
    pi = models.f3x4_codon_freqs(POS_FREQS)
    tmp = tempfile.mkdtemp(prefix="hygold_")
    fasta, outp, supp = (os.path.join(tmp, n) for n in ("aln.fasta", "out.txt", "support.txt"))
    hbl.write_fasta(fasta, flat.leaf_names, syn.seqs)
    txt = hbl.build_script(fasta=fasta, newick=tree.to_newick(syn.tree), unit=3, model_block=hbl.codon_model_block(tmpl, pi),
                           model_name="MGM", globals_=dict(R=0.3, **REV), branch_t=bt, out_path=outp, per_site=False)
    txt += ("DataSet anc = ReconstructAncestors (lf, MARGINAL);\n"
            "m_ = anc.marginal_support_matrix;\n"
            f'fprintf ("{supp}", CLEAR_FILE, Rows (m_), " ", Columns (m_), "\\n");\n'
            f'for (i_ = 0; i_ < Rows (m_); i_ += 1) {{ for (j_ = 0; j_ < Columns (m_); j_ += 1) {{ fprintf ("{supp}", Format (m_[i_][j_], 24, 17), "\\n"); }} }}\n')
    hbl.run_script(txt, tmp)
    with open(supp) as fh:
        r, c = (int(x) for x in fh.readline().split())
        vals = np.array([float(x) for x in fh.read().split()]).reshape(r, c)
    pd = data.compress(syn.seqs, 3)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), support=vals, D=61, L=flat.L, flat_parents=flat.flat_parents,
                        leaf_codes=pd.leaf_codes, ambig=pd.ambig, pattern_freq=pd.pattern_freq,
                        t=np.array([bt[n] for n in flat.branch_names()]), omega=0.3,
                        rev=np.array([REV[k] for k in ("AC", "AT", "CG", "CT", "GT")]), pos_freqs=POS_FREQS, root_freqs=pi,
                        kind="codon")
    print(f"{name}: support matrix {vals.shape}, row sums (first pattern) {vals[:, :61].sum(1)[:3]}")


def reference_test_case(name="ref_smallcodon"):
    """The reference's OWN known-answer test for this path: tests/hbltests/SimpleOptimizations/SmallCodon.bf — an HIV-1 RT
    alignment (8 sequences x 440 codons) fitted with MG94x012232 (AC, AT = CG = GT, CT free; AG = 1), position-specific
    frequency constants, expected maximised log L = -3189.516375 (`_expectedLL`, :37; tolerance of the reference's
    harness: 2 x OPTIMIZATION_PRECISION).  The fixture takes DATA from that file (alignment, tree, expected value); the
    model is rebuilt by our own generator (the test file's constants are the 6-digit position frequencies of the
    alignment, which we recompute).  Stored: everything the other codon fixtures have, evaluated at the test's start
    point (all rates 1, branch parameters 0.1), plus the fit of the unmodified binary driven by OUR script."""
    import re
    path = "/root/reference/tests/hbltests/SimpleOptimizations/SmallCodon.bf"
    txt = open(path).read()
    block = re.search(r"MATRIX\s*(.*?)\nEND;", txt, re.S).group(1)
    names, seqs = [], []
    for row in block.strip().splitlines():
        m = re.match(r"\s*'([^']+)'\s+([ACGT]+)", row)
        if m:
            names.append(m.group(1))
            seqs.append(m.group(2))
    newick = re.search(r"TREE tree = (.*?);", txt).group(1)
    expected = float(re.search(r"_expectedLL\s*=\s*(-?[0-9.]+)", txt).group(1))
    cnt = np.zeros((3, 4))
    for sq in seqs:
        for i, ch in enumerate(sq):
            cnt[i % 3, "ACGT".index(ch)] += 1
    pf = cnt / cnt.sum(1, keepdims=True)
    pf_model = np.array([[float("%.6g" % v) for v in row] for row in pf])   # the constants as the test file prints them
    root = tree.parse_newick(newick + ";")
    flat = tree.flatten(root)
    order = [names.index(n) for n in flat.leaf_names]
    seqs_flat = [seqs[k] for k in order]
    bt = {n: 0.1 for n in flat.branch_names()}
    pi = models.f3x4_codon_freqs(pf)
    g = dict(R=1.0, AC=1.0, AT=1.0, CG=1.0, CT=1.0, GT=1.0)
    common_args = dict(names=flat.leaf_names, seqs=seqs_flat, newick=tree.to_newick(root), unit=3,
                       model_block=hbl.codon_model_block(models.mg94rev_template(pf_model), pi), model_name="MGM",
                       globals_=g, branch_t=bt, constraints=dict(CG="AT", GT="AT"))
    res = hbl.evaluate(**common_args)                                   # start point: log L + per-site log L
    res["opt_logl"] = hbl.evaluate(optimize=True, per_site=False, **common_args)["opt_logl"]   # the fit
    pd = data.compress(seqs_flat, 3)
    fx = dict(kind="codon", D=61, L=flat.L, flat_parents=flat.flat_parents, leaf_codes=pd.leaf_codes,
              ambig=pd.ambig, pattern_freq=pd.pattern_freq, site_to_pattern=pd.site_to_pattern,
              t=np.array([bt[n] for n in flat.branch_names()]), omega=1.0,
              rev=np.array([1.0, 1.0, 1.0, 1.0, 1.0]), pos_freqs=pf_model, root_freqs=pi, logl=res["logl"],
              site_logl=res["site_logl"], names=np.array(flat.leaf_names), seqs=np.array(seqs_flat),
              newick=np.array(tree.to_newick(root)), expected_opt_logl=expected, ref_opt_logl=res["opt_logl"])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print(f"{name}: start logL = {res['logl']!r}  fitted = {res['opt_logl']!r}  reference test expects {expected!r}  S = {pd.S}")


def reference_test_case_nuc(name="ref_fluHA"):
    """Real data for the 4-state path from the reference's test tests/hbltests/SimpleOptimizations/IntermediateNuc.bf: HKY85
    on an Influenza A HA alignment, 349 sequences x 967 nucleotides (tests/hbltests/data/fluHA.nex).  DATA taken from the
    reference: alignment, tree with its branch lengths, the model's equilibrium frequencies; the model block is our
    generator's (HKY85 = REV with AC = AT = CG = GT = TVTS, CT = 1).  Evaluated at TVTS = 0.25 (the test's start) with
    branch parameters = the tree file's lengths (floored at 1e-4).  Not used as an optimisation known answer: the test's
    `_expectedLL` (-11389.4544) is not reproduced by the reference itself (its own script ends at -11389.8206 with this
    build and fails its assertion; the default optimiser started from the tree's lengths reaches -11369.5648)."""
    import re
    base = "/root/reference/tests/hbltests"
    bf = open(os.path.join(base, "SimpleOptimizations", "IntermediateNuc.bf")).read()
    freqs = np.array([float(x) for x in re.findall(r"\{\s*([0-9.]+)\}", bf[bf.index("flu_part2_Freqs"):bf.index("Model flu_part2")])])
    assert freqs.shape == (4,) and abs(freqs.sum() - 1) < 1e-9
    txt = open(os.path.join(base, "data", "fluHA.nex")).read()
    block = txt[txt.index("MATRIX") + 6: txt.index("END;", txt.index("MATRIX"))]
    names, seqs = [], []
    for row in block.splitlines():
        m = re.match(r"\s*'([^']+)'\s+([ACGT]+)", row)
        if m:
            names.append(m.group(1))
            seqs.append(m.group(2))
    newick = re.search(r"TREE tree = (.*?);", txt, re.S).group(1)
    root = tree.parse_newick(newick + ";")
    flat = tree.flatten(root)
    seqs_flat = [seqs[names.index(n)] for n in flat.leaf_names]
    lengths = {}

    def walk(n):
        for c in n.children:
            walk(c)
        if n.parent is not None:
            lengths[n.name] = max(float(n.length or 0.0), 1e-4)
    walk(root)
    bt = {n: lengths[n] for n in flat.branch_names()}
    g = dict(TVTS=0.25, AC=0.25, AT=0.25, CG=0.25, CT=1.0, GT=0.25)
    cons = dict(AC="TVTS", AT="TVTS", CG="TVTS", GT="TVTS", CT="1")
    common_args = dict(names=flat.leaf_names, seqs=seqs_flat, newick=tree.to_newick(root), unit=1,
                       model_block=hbl.nuc_model_block(freqs), model_name="NM", globals_=g, branch_t=bt, constraints=cons)
    res = hbl.evaluate(**common_args)
    pd = data.compress(seqs_flat, 1)
    fx = dict(kind="nuc", D=4, L=flat.L, flat_parents=flat.flat_parents, leaf_codes=pd.leaf_codes,
              ambig=pd.ambig, pattern_freq=pd.pattern_freq, site_to_pattern=pd.site_to_pattern,
              t=np.array([bt[n] for n in flat.branch_names()]), rev=np.array([0.25, 0.25, 0.25, 1.0, 0.25]),
              root_freqs=freqs, logl=res["logl"], site_logl=res["site_logl"], names=np.array(flat.leaf_names),
              seqs=np.array(seqs_flat), newick=np.array(tree.to_newick(root)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print(f"{name}: logL = {res['logl']!r}  S = {pd.S}")


# ---------------------------------------------------------------------------------------------------------------
# BASELINE.json configurations at their STATED sizes ("no -m gpu test runs any BASELINE config at its
# stated size").  The alignments are the synthetic ones bench.py uses (hyphy_amd.data.evolve with bench.py's seeds:
# regenerated from the seed on the GPU box, checked against the CRC stored here), evaluated by the unmodified
# reference at bench.py's first parameter point (all branch parameters 0.05, omega 0.3).  Stored: scalar log L and
# per-site log L (all sites up to 10 000 codons, 2 000 sampled sites beyond).
#     python -m oracle.make_golden fullsize
# ---------------------------------------------------------------------------------------------------------------
def fel_case(name="ref_fel_12x60", n_taxa=12, n_codons=60, seed=91, branches="Internal", threads=8, analysis="FEL"):
    """The reference's OWN FEL analysis (res/TemplateBatchFiles/SelectionAnalyses/FEL.bf, unmodified, run by the unmodified
    binary) on a synthetic alignment: nucleotide GTR fit -> global MG94xREV fit -> one single-site likelihood function per
    site, alpha / beta_test / beta_nuisance optimised under the alternative and under beta_test := alpha, LRT and p-value
    (FEL.bf:593-1000).  A two-line wrapper executes FEL.bf and then prints `fel.final_partitioned_mg_results` (the global
    fit the site phase starts from: per-branch synonymous rates, the theta's, omega) next to FEL's own JSON.
    The fixture holds the inputs of the site phase (tree, per-site codon states, which branches are tested, the global
    MLEs, position frequencies recovered from the CF3x4 vector: the product form is exact) and the reference's per-site
    table [alpha, beta, alpha=beta, LRT, p-value]; tests/test_gpu_parity.py::test_fel_driver_matches_the_reference_fel
    runs hyphy_amd/fel.py on the device against it.  The reconstructed global model is checked here against the reference's
    global log-likelihood with the CPU oracle before anything is written.
    analysis="MEME": the same for MEME.bf (per-site two-class branch-site mixture on the tested branches; table columns
    alpha, beta-, p-, beta+, p+, LRT, p-value, MEME LogL, FEL LogL — the last two are the site's log-likelihoods at the
    reference's optima: a likelihood-level anchor for hyphy_amd/fel.py::meme)."""
    import json
    import re
    import subprocess
    import tempfile
    from oracle import oracle
    syn = data.evolve(n_taxa, n_codons, 3, seed=seed, p_change=0.10)
    flat = syn.flat
    tmp = tempfile.mkdtemp(prefix="felref_")
    hbl.write_fasta(os.path.join(tmp, "aln.fasta"), flat.leaf_names, syn.seqs)
    with open(os.path.join(tmp, "tree.nwk"), "w") as fh:
        fh.write(tree.to_newick(syn.tree) + ";\n")
    fel_bf = f"/root/reference/res/TemplateBatchFiles/SelectionAnalyses/{analysis}.bf"
    with open(os.path.join(tmp, "wrap.bf"), "w") as fh:
        fh.write(f'ExecuteAFile ("{fel_bf}");\nfprintf ("{tmp}/dump.txt", CLEAR_FILE, {analysis.lower()}.final_partitioned_mg_results);\n')
    r = subprocess.run([hbl.REF_BIN, "LIBPATH=/root/reference/res", f"CPU={threads}", os.path.join(tmp, "wrap.bf"),
                        "--alignment", os.path.join(tmp, "aln.fasta"), "--tree", os.path.join(tmp, "tree.nwk"), "--code", "Universal",
                        "--branches", branches] + (["--srv", "Yes", "--ci", "No"] if analysis == "FEL" else []) +
                       ["--pvalue", "0.1", "--resample", "0", "--output", os.path.join(tmp, "fel.json")], capture_output=True, text=True, timeout=3600)
    if r.returncode != 0:
        raise RuntimeError(r.stdout[-3000:] + r.stderr[-2000:])
    d = open(os.path.join(tmp, "dump.txt")).read()
    num = r"([-0-9.e+]+)"
    logl = float(re.search(r'"LogL":' + num, d).group(1))
    syn_rate = {m.group(1): float(m.group(2)) for m in re.finditer(
        r'"(\w+)":\{\s*"MLE":[-0-9.e+]+,\s*"non-synonymous rate":\{.*?\},\s*"synonymous rate":\{\s*"ID":"alpha",\s*"MLE":' + num, d, re.S)}
    total_len = {m.group(1): float(m.group(2)) for m in re.finditer(r'"(\w+)":\{\s*"MLE":' + num + r',\s*"non-synonymous rate"', d)}
    theta = {m.group(1) + m.group(2): float(m.group(3)) for m in re.finditer(
        r'"Substitution rate from nucleotide (\w) to nucleotide (\w)":\{\s*"ID":"[^"]+",\s*"MLE":' + num, d)}
    om_t = float(re.search(r'rate ratio for \*test\*":\{\s*"ID":"[^"]+",\s*"MLE":' + num, d).group(1))
    m_bg = re.search(r'rate ratio for \*background\*":\{\s*"ID":"[^"]+",\s*"MLE":' + num, d)
    om_b = float(m_bg.group(1)) if m_bg else om_t
    j = json.load(open(os.path.join(tmp, "fel.json")))
    efv = np.array(j["fits"]["Global MG94xREV"]["Equilibrium frequencies"]).ravel()
    A = np.zeros((61, 13))
    for row, c in enumerate(models.sense_codons()):
        for pos in range(3):
            A[row, 4 * pos + "ACGT".index(c[pos])] = 1.0
    A[:, 12] = 1.0
    x = np.linalg.lstsq(A, np.log(efv), rcond=None)[0]
    pf = np.exp(x[:12]).reshape(3, 4)
    pf /= pf.sum(1, keepdims=True)
    pi = models.f3x4_codon_freqs(pf)
    assert np.abs(pi - efv).max() < 1e-14, "CF3x4 vector is not of product form?"
    names = flat.branch_names()
    ts = np.array([syn_rate[n] for n in names])
    tested = np.array([j["tested"]["0"][n] == "test" for n in names])
    rev = np.array([theta["AC"], theta["AT"], theta["CG"], theta["CT"], theta["GT"]])
    revd = dict(AC=rev[0], AT=rev[1], CG=rev[2], CT=rev[3], GT=rev[4])
    assert theta["AG"] == 1.0
    Q = np.stack([models.mg94rev_Q(ts[b], om_t if tested[b] else om_b, revd, pf) for b in range(len(names))])
    pd = data.compress(syn.seqs, 3)
    op = oracle.OraclePartition(61, flat.flat_parents, flat.L, pd.leaf_codes, pd.ambig, pd.pattern_freq)
    nodes = np.arange(len(names), dtype=np.int64)
    op.set_P(nodes, oracle.expm(Q, True))
    mine = op.compute_block(nodes, pi)
    assert abs(mine - logl) <= 1e-10 * abs(logl), (mine, logl)   # the reconstructed global model IS the reference's
    full = np.array(j["MLE"]["content"]["0"], dtype=np.float64)
    heads = ["alpha", "beta", "alpha=beta", "LRT", "p-value"]
    table = full[:, :5]
    if analysis == "MEME":
        heads = ["alpha", "beta-", "p-", "beta+", "p+", "LRT", "p-value", "MEME LogL", "FEL LogL"]
        assert "MEME LogL" in j["MLE"]["headers"][9][0] and "FEL LogL" in j["MLE"]["headers"][10][0]
        table = full[:, [0, 1, 2, 3, 4, 5, 6, 9, 10]]
    assert table.shape[0] == n_codons
    sites = data.from_states(syn.states, 61, compress_patterns=False)
    fx = dict(kind="codon", D=61, L=flat.L, flat_parents=flat.flat_parents, leaf_codes=sites.leaf_codes, tested=tested,
              syn_rate=ts, branch_length=np.array([total_len[n] for n in names]), rev=rev, omega_test=om_t, omega_background=om_b, pos_freqs=pf, root_freqs=pi, global_logl=logl,
              fel_table=table, fel_headers=np.array(heads), analysis=np.array(analysis),
              names=np.array(flat.leaf_names), seqs=np.array(syn.seqs), newick=np.array(tree.to_newick(syn.tree)),
              branches=np.array(branches))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print(f"{name}: global logL {logl!r} (oracle on the reconstructed model: {mine!r}); {int(tested.sum())} of {len(names)} branches tested; "
          f"{int((table[:, 6 if analysis == 'MEME' else 4] <= 0.1).sum())} sites at p <= 0.1")


def _simulate_site_classes(flat, t, omegas, weights, rev, pf, n_codons, seed):
    """Codon alignment evolved on `flat` under MG94xREV with a per-SITE omega class (our own sampler: root state from the
    CF3x4 vector, child state from the row of P_b^(class) — enough signal for BUSTED to infer a non-degenerate mixture)."""
    from oracle import oracle
    rng = np.random.default_rng(seed)
    pi = models.f3x4_codon_freqs(pf)
    B = flat.n_branches
    P = [oracle.expm(np.stack([models.mg94rev_Q(t[b], om, rev, pf) for b in range(B)]), True) for om in omegas]
    n_nodes = B + 1
    states = np.zeros((n_nodes, n_codons), dtype=np.int64)
    cls = rng.choice(len(omegas), size=n_codons, p=np.asarray(weights))
    root = n_nodes - 1
    states[root] = rng.choice(61, size=n_codons, p=pi / pi.sum())
    for node in range(B - 1, -1, -1):          # (parents have larger flat indices than their children)
        par = flat.L + int(flat.flat_parents[node])
        for s_ in range(n_codons):
            row = np.maximum(P[cls[s_]][node][states[par, s_]], 0.0)
            states[node, s_] = rng.choice(61, p=row / row.sum())
    cod = models.sense_codons()
    return ["".join(cod[k] for k in states[leaf]) for leaf in range(flat.L)], states[:flat.L]


def busted_case(name="ref_busted_16x150", n_taxa=16, n_codons=150, seed=131, branches="Internal", threads=8):
    """The reference's OWN BUSTED analysis (res/TemplateBatchFiles/SelectionAnalyses/BUSTED.bf, unmodified, unmodified binary;
    --srv No, 3 omega classes) on an alignment simulated with site classes omega = 0.1 / 1 / 8.  A two-line wrapper executes
    BUSTED.bf and prints `busted.full_model`: the MLEs of the unconstrained branch-site model — per-branch t, theta's, and for
    the test and the background set three omegas with their stick-breaking weights.  Fixture: alignment + tree + those MLEs +
    the reference's log L at them.  The model is the explicit-form mixture P_b = sum_k w_k Exp(Q_b(omega_k))
    (libv3/models/codon/BS_REL.bf) — the CPU oracle must reproduce the log L from the reconstruction before the fixture is
    written; tests/test_gpu_parity.py::test_busted_fit_of_the_reference_evaluates_on_the_device holds
    hyphy_hip_evaluate_mixture to it."""
    import json
    import re
    import subprocess
    import tempfile
    from oracle import oracle
    rng = np.random.default_rng(seed)
    root = tree.random_tree(n_taxa, rng, trifurcating_root=True)
    flat = tree.flatten(root)
    t_sim = rng.uniform(0.05, 0.3, flat.n_branches)
    seqs, _ = _simulate_site_classes(flat, t_sim, (0.1, 1.0, 8.0), (0.5, 0.35, 0.15), REV, POS_FREQS, n_codons, seed + 1)
    tmp = tempfile.mkdtemp(prefix="bustedref_")
    hbl.write_fasta(os.path.join(tmp, "aln.fasta"), flat.leaf_names, seqs)
    with open(os.path.join(tmp, "tree.nwk"), "w") as fh:
        fh.write(tree.to_newick(root) + ";\n")
    bf = "/root/reference/res/TemplateBatchFiles/SelectionAnalyses/BUSTED.bf"
    with open(os.path.join(tmp, "wrap.bf"), "w") as fh:
        fh.write(f'ExecuteAFile ("{bf}");\nfprintf ("{tmp}/dump.txt", CLEAR_FILE, busted.full_model);\n')
    r = subprocess.run([hbl.REF_BIN, "LIBPATH=/root/reference/res", f"CPU={threads}", os.path.join(tmp, "wrap.bf"),
                        "--alignment", os.path.join(tmp, "aln.fasta"), "--tree", os.path.join(tmp, "tree.nwk"), "--code", "Universal",
                        "--branches", branches, "--srv", "No", "--rates", "3", "--starting-points", "1",
                        "--output", os.path.join(tmp, "busted.json")], capture_output=True, text=True, timeout=7200)
    if r.returncode != 0:
        raise RuntimeError(r.stdout[-3000:] + r.stderr[-2000:])
    d = open(os.path.join(tmp, "dump.txt")).read()
    num = r"([-0-9.e+]+)"
    logl = float(re.search(r'"LogL":' + num, d).group(1))
    tb = {m.group(1): float(m.group(2)) for m in re.finditer(r'"(\w+)":\{\s*"MLE":[-0-9.e+]+,\s*"synonymous rate":\{\s*"ID":"t",\s*"MLE":' + num, d, re.S)}
    def g(idname):
        return float(re.search(r'"ID":"' + re.escape(idname) + r'",\s*"MLE":' + num, d).group(1))
    rev = np.array([g("busted.test.theta_" + k) for k in ("AC", "AT", "CG", "CT", "GT")])
    om = {st: np.array([g(f"busted.{st}.omega{k}") for k in (1, 2, 3)]) for st in ("test", "background")}
    w = {}
    for st in ("test", "background"):
        a0, a1 = g(f"busted.{st}.bsrel_mixture_aux_0"), g(f"busted.{st}.bsrel_mixture_aux_1")
        w[st] = np.array([a0, (1.0 - a0) * a1, (1.0 - a0) * (1.0 - a1)])     # stick breaking (BS_REL.bf)
    j = json.load(open(os.path.join(tmp, "busted.json")))
    efv = np.array(j["fits"]["MG94xREV with separate rates for branch sets"]["Equilibrium frequencies"]).ravel()   # (CF3x4: the same vector for every codon model of the run)
    A = np.zeros((61, 13))
    for row, c in enumerate(models.sense_codons()):
        for pos in range(3):
            A[row, 4 * pos + "ACGT".index(c[pos])] = 1.0
    A[:, 12] = 1.0
    pf = np.exp(np.linalg.lstsq(A, np.log(efv), rcond=None)[0][:12]).reshape(3, 4)
    pf /= pf.sum(1, keepdims=True)
    pi = models.f3x4_codon_freqs(pf)
    assert np.abs(pi - efv).max() < 1e-14
    names = flat.branch_names()
    ts = np.array([tb[n] for n in names])
    tested = np.array([j["tested"]["0"][n] == "test" for n in names])
    revd = dict(zip(("AC", "AT", "CG", "CT", "GT"), rev))
    B = len(names)
    Qc = np.zeros((B, 3, 61, 61))
    W = np.zeros((B, 3))
    for b in range(B):
        st = "test" if tested[b] else "background"
        W[b] = w[st]
        for k in range(3):
            Qc[b, k] = models.mg94rev_Q(ts[b], om[st][k], revd, pf)
    pd = data.compress(seqs, 3)
    op = oracle.OraclePartition(61, flat.flat_parents, flat.L, pd.leaf_codes, pd.ambig, pd.pattern_freq)
    nodes = np.arange(B, dtype=np.int64)
    Pm = sum(W[:, k, None, None] * oracle.expm(Qc[:, k], True) for k in range(3))
    op.set_P(nodes, Pm)
    mine = op.compute_block(nodes, pi)
    assert abs(mine - logl) <= 1e-9 * abs(logl), (mine, logl)
    fits = {k: float(v["Log Likelihood"]) for k, v in j["fits"].items()}
    fx = dict(kind="codon_mixture", D=61, L=flat.L, flat_parents=flat.flat_parents, leaf_codes=pd.leaf_codes, ambig=pd.ambig,
              pattern_freq=pd.pattern_freq, site_to_pattern=pd.site_to_pattern, tested=tested, t=ts, rev=rev,
              omega_test=om["test"], omega_background=om["background"], weights_test=w["test"], weights_background=w["background"],
              pos_freqs=pf, root_freqs=pi, logl=logl, json_unconstrained_logl=fits.get("Unconstrained model", np.nan),
              json_constrained_logl=fits.get("Constrained model", np.nan), p_value=float(j["test results"]["p-value"]),
              names=np.array(flat.leaf_names), seqs=np.array(seqs), newick=np.array(tree.to_newick(root)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print(f"{name}: unconstrained logL {logl!r} (oracle on the reconstructed mixture: {mine!r}); test omegas {om['test']} weights {w['test']}; "
          f"background omegas {om['background']} weights {w['background']}; BUSTED p = {fx['p_value']}")


def _hbl_matrix(text, key_pos):
    """The numeric matrix printed by HBL behind position `key_pos` of `text` ("{ {a, b} {c, d} }" with arbitrary white space)."""
    i = text.index("{", key_pos)
    depth, j = 0, i
    while True:
        ch = text[j]
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
            if depth == 0:
                break
        j += 1
    body = text[i + 1:j]
    rows = [r for r in body.replace("\n", " ").split("}") if "{" in r]
    return np.array([[float(x) for x in r.split("{")[1].split(",")] for r in rows])


def fubar_case(name="ref_fubar_12x60", n_taxa=12, n_codons=60, seed=91, grid=10, threads=8):
    """The reference's OWN FUBAR analysis (res/TemplateBatchFiles/SelectionAnalyses/FUBAR.bf, unmodified, unmodified binary): its
    grid phase evaluates the site log-likelihoods of the alignment under MG94xREV at every point (alpha, beta) of a grid x grid
    rate grid — site rates x per-branch (synonymous, non-synonymous) factors taken from the nucleotide GTR fit — which is exactly
    the shape of hyphy_hip_site_fits_evaluate (one parameter set per grid point).  FUBAR's cache file keeps that matrix as per-site
    softmax columns + log normalisers (modules/grid_compute.ibf: ConvertToConditionals), so the raw site log-likelihoods are
    recovered as log(conditional) + scaler.  A two-line wrapper also prints `fubar.pass1.mle` (the per-branch factors, theta's).
    Fixture: per-site codon states, tree, factors, theta's, position frequencies, the grid, and the reference's [grid point][site]
    log-likelihood matrix; a sample of entries is checked against the CPU oracle before anything is written."""
    import json
    import re
    import subprocess
    import tempfile
    from oracle import oracle
    syn = data.evolve(n_taxa, n_codons, 3, seed=seed, p_change=0.10)
    flat = syn.flat
    tmp = tempfile.mkdtemp(prefix="fubarref_")
    hbl.write_fasta(os.path.join(tmp, "aln.fasta"), flat.leaf_names, syn.seqs)
    with open(os.path.join(tmp, "tree.nwk"), "w") as fh:
        fh.write(tree.to_newick(syn.tree) + ";\n")
    bf = "/root/reference/res/TemplateBatchFiles/SelectionAnalyses/FUBAR.bf"
    with open(os.path.join(tmp, "wrap.bf"), "w") as fh:
        # (the factors are read back AFTER the run, from the constraints the grid phase left on the tree: FUBAR re-applies its first-pass
        #  estimates through the model's set-branch-length hook, a root finder whose answer differs from the values it printed
        #  before by a few 1e-7 relative)
        fh.write(f'ExecuteAFile ("{bf}");\nfubar.after = estimators.ExtractMLEs ("fubar.lf.codon", fubar.model_id_to_object);\n'
                 f'fprintf ("{tmp}/dump.txt", CLEAR_FILE, fubar.after);\n')
    r = subprocess.run([hbl.REF_BIN, "LIBPATH=/root/reference/res", f"CPU={threads}", os.path.join(tmp, "wrap.bf"),
                        "--alignment", os.path.join(tmp, "aln.fasta"), "--tree", os.path.join(tmp, "tree.nwk"), "--code", "Universal",
                        "--grid", str(grid), "--method", "Variational-Bayes", "--cache", os.path.join(tmp, "fubar.cache"),
                        "--output", os.path.join(tmp, "fubar.json")], capture_output=True, text=True, timeout=3600)
    if r.returncode != 0:
        raise RuntimeError(r.stdout[-3000:] + r.stderr[-2000:])
    d = open(os.path.join(tmp, "dump.txt")).read()
    c = open(os.path.join(tmp, "fubar.cache")).read()
    num = r"([-0-9.e+]+)"
    alpha_b = {m.group(1): float(m.group(2)) for m in re.finditer(
        r'"(\w+)":\{\s*"MLE":[-0-9.e+]+,\s*"non-synonymous rate":\{.*?\},\s*"synonymous rate":\{\s*"ID":"alpha",\s*"MLE":[-0-9.e+]+,\s*'
        r'"constraint":"fubar\.scaler\.alpha\*' + num + '"', d, re.S)}
    beta_b = {m.group(1): float(m.group(2)) for m in re.finditer(
        r'"(\w+)":\{\s*"MLE":[-0-9.e+]+,\s*"non-synonymous rate":\{\s*"ID":"beta",\s*"MLE":[-0-9.e+]+,\s*"constraint":"fubar\.scaler\.beta\*' + num + '"', d, re.S)}
    assert len(alpha_b) == flat.n_branches and len(beta_b) == flat.n_branches, (len(alpha_b), len(beta_b))
    theta = {m.group(1) + m.group(2): float(m.group(3)) for m in re.finditer(
        r'"Substitution rate from nucleotide (\w) to nucleotide (\w)":\{\s*"ID":"[^"]+",\s*"MLE":' + num, d)}
    efv = _hbl_matrix(d, d.index("{", d.index('"EFV"')) + 1).ravel()      # ("EFV": {"<model id>": <matrix>})
    assert efv.shape == (61,), efv.shape
    A = np.zeros((61, 13))
    for row, cdn in enumerate(models.sense_codons()):
        for pos in range(3):
            A[row, 4 * pos + "ACGT".index(cdn[pos])] = 1.0
    A[:, 12] = 1.0
    pf = np.exp(np.linalg.lstsq(A, np.log(efv), rcond=None)[0][:12]).reshape(3, 4)
    pf /= pf.sum(1, keepdims=True)
    pi = models.f3x4_codon_freqs(pf)
    assert np.abs(pi - efv).max() < 1e-14
    cond = _hbl_matrix(c, c.index('"conditionals"', c.index('"conditionals"') + 5))   # [grid points][sites]
    scal = _hbl_matrix(c, c.index('"scalers"')).ravel()
    G = grid * grid
    assert cond.shape == (G, n_codons) and scal.shape == (n_codons,), (cond.shape, scal.shape)
    with np.errstate(divide="ignore"):
        site_logl = np.log(cond) + scal[None, :]          # (-inf where the reference's softmax underflowed to 0)
    j = json.load(open(os.path.join(tmp, "fubar.json")))
    gridm = np.array(j["grid"], dtype=np.float64)[:, :2]
    assert gridm.shape == (G, 2)
    names = flat.branch_names()
    ca = np.array([alpha_b[n] for n in names])
    cb = np.array([beta_b[n] for n in names])
    rev = np.array([theta["AC"], theta["AT"], theta["CG"], theta["CT"], theta["GT"]])
    assert theta["AG"] == 1.0
    sites = data.from_states(syn.states, 61, compress_patterns=False)
    # a sample of (grid point, site) entries against the CPU oracle (explicit exponentials, one pattern)
    T = np.zeros((2, 61, 61))
    rv = dict(zip(("AC", "AT", "CG", "CT", "GT"), rev), AG=1.0)
    for (i, jj, nm, ns, f) in models.mg94rev_template(pf):
        T[1 if ns else 0, i, jj] = rv[nm] * f
    nodes = np.arange(len(names), dtype=np.int64)
    rng = np.random.default_rng(1)
    worst = 0.0
    for _ in range(40):
        g, st = int(rng.integers(G)), int(rng.integers(n_codons))
        if not np.isfinite(site_logl[g, st]) or cond[g, st] < 1e-12:
            continue
        Q = np.stack([gridm[g, 0] * ca[b] * T[0] + gridm[g, 1] * cb[b] * T[1] for b in range(len(names))])
        for b in range(len(names)):
            np.fill_diagonal(Q[b], 0.0)
            np.fill_diagonal(Q[b], -Q[b].sum(1))
        op = oracle.OraclePartition(61, flat.flat_parents, flat.L, sites.leaf_codes[:, st:st + 1], None, np.ones(1))
        op.set_P(nodes, oracle.expm(Q, True))
        mine = op.compute_block(nodes, pi)
        worst = max(worst, abs(mine - site_logl[g, st]) / abs(mine))
    assert worst < 1e-9, worst
    fx = dict(kind="codon", D=61, L=flat.L, flat_parents=flat.flat_parents, leaf_codes=sites.leaf_codes, syn_factor=ca, nonsyn_factor=cb,
              rev=rev, pos_freqs=pf, root_freqs=pi, grid=gridm, site_logl=site_logl, conditionals=cond, scalers=scal,
              names=np.array(flat.leaf_names), seqs=np.array(syn.seqs), newick=np.array(tree.to_newick(syn.tree)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print(f"{name}: {G} grid points x {n_codons} sites; {int(np.isfinite(site_logl).sum())} finite entries; "
          f"sampled entries agree with the oracle to {worst:.1e}")


def _crc(a):
    import zlib
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff


def _site_sample(n_sites, seed):
    if n_sites <= 10000:
        return np.arange(n_sites)
    return np.sort(np.random.default_rng(seed).choice(n_sites, size=2000, replace=False))


def fullsize_codon(name, taxa, codons, seed, classes=None, threads=8):
    syn = data.evolve(taxa, codons, 3, seed=seed, p_change=0.04)
    flat = syn.flat
    bt = {n: 0.05 for n in flat.branch_names()}
    tmpl = models.mg94rev_template(POS_FREQS)
    pi = models.f3x4_codon_freqs(POS_FREQS)
    g = dict(R=0.3, **REV)
    category = None
    omega_expr = "R"
    if classes:   # BUSTED-style: 3 omega classes mixed per site (weighted-sum category mode), omega_c = R * cc
        category = dict(name="cc", weights=classes["weights"], values=classes["values"])
        omega_expr = "R*cc"
    res = hbl.evaluate(names=flat.leaf_names, seqs=syn.seqs, newick=tree.to_newick(syn.tree), unit=3,
                       model_block=hbl.codon_model_block(tmpl, pi, omega=omega_expr), model_name="MGM", globals_=g, branch_t=bt,
                       category=category, threads=threads, timeout=3600.0)
    idx = _site_sample(codons, seed)
    fx = dict(kind="codon_full", taxa=taxa, sites=codons, seed=seed, p_change=0.04, states_crc=_crc(syn.states.astype(np.int16)),
              t=0.05, omega=0.3, rev=np.array([REV[k] for k in ("AC", "AT", "CG", "CT", "GT")]), pos_freqs=POS_FREQS,
              logl=res["logl"], site_index=idx, site_logl=res["site_logl"][idx])
    if classes:
        fx["cat_weights"] = np.array(classes["weights"])
        fx["cat_values"] = np.array(classes["values"])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print(f"{name}: logL = {res['logl']!r}  sites stored {len(idx)}")


def fullsize_mix3(name="full_mix3_64x10k", taxa=64, codons=10000, seed=3, omegas=(0.1, 1.0, 5.0), weights=(0.6, 0.3, 0.1), threads=16):
    """configs[2] in the form BUSTED really runs (BS_REL.bf:48, tree.cpp:3047-3090): every branch's transition matrix is the
    explicit-form mixture sum_m w_m Exp(Q_b(omega_m)) over three omega classes — bench.py's headline alignment, branch lengths 0.05,
    the weights / omegas of tools/adapter_rate.py's `mix3` rows.  Scalar log L + 2 000 sampled per-site values."""
    syn = data.evolve(taxa, codons, 3, seed=seed, p_change=0.04)
    flat = syn.flat
    bt = {n: 0.05 for n in flat.branch_names()}
    tmpl = models.mg94rev_template(POS_FREQS)
    pi = models.f3x4_codon_freqs(POS_FREQS)
    M = len(omegas)
    g = dict(REV)
    for m, om in enumerate(omegas, start=1):
        g[f"R{m}"] = om
    for m, w in enumerate(weights[:-1], start=1):
        g[f"W{m}"] = w
    wexpr = [f"W{m}" for m in range(1, M)] + ["(1" + "".join(f"-W{m}" for m in range(1, M)) + ")"]
    block = hbl.codon_mixture_model_block(tmpl, pi, [f"R{m}" for m in range(1, M + 1)], wexpr)
    res = hbl.evaluate(names=flat.leaf_names, seqs=syn.seqs, newick=tree.to_newick(syn.tree), unit=3, model_block=block,
                       model_name="MGM", globals_=g, branch_t=bt, upper_bounds={f"W{m}": 1.0 for m in range(1, M)}, threads=threads,
                       timeout=3600.0)
    idx = np.sort(np.random.default_rng(seed).choice(codons, size=2000, replace=False))
    fx = dict(kind="codon_mixture_full", taxa=taxa, sites=codons, seed=seed, p_change=0.04, states_crc=_crc(syn.states.astype(np.int16)),
              t=0.05, omegas=np.array(omegas), weights=np.array(weights), rev=np.array([REV[k] for k in ("AC", "AT", "CG", "CT", "GT")]),
              pos_freqs=POS_FREQS, logl=res["logl"], site_index=idx, site_logl=res["site_logl"][idx])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print(f"{name}: logL = {res['logl']!r}  sites stored {len(idx)}")


def fullsize_nuc(name="full_hky_8x1k", taxa=8, sites=1000, seed=1, kappa=0.35, threads=1):
    """configs[0]: HKY85 (transversions kappa x transitions, the parameterisation of the reference's IntermediateNuc.bf), 8 taxa x
    1 000 sites — bench.py's `hky_8x1k` alignment at its stated size: scalar and per-site log L of the reference, and its log L at
    the first sweep points of a global (kappa' = kappa + 0.001 k on the four transversion rates) as the adapter test uses them."""
    syn = data.evolve(taxa, sites, 1, seed=seed, p_change=0.04)
    flat = syn.flat
    bt = {n: 0.05 for n in flat.branch_names()}
    rev = models.hky85_rev(kappa)
    res = hbl.evaluate(names=flat.leaf_names, seqs=syn.seqs, newick=tree.to_newick(syn.tree), unit=1,
                       model_block=hbl.nuc_model_block(NUC_FREQS), model_name="NM", globals_=rev, branch_t=bt,
                       threads=threads, timeout=600.0)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="nuc_full", taxa=taxa, sites=sites, seed=seed, p_change=0.04,
                        states_crc=_crc(syn.states.astype(np.int16)), t=0.05, kappa=kappa,
                        rev=np.array([rev[k] for k in ("AC", "AT", "CG", "CT", "GT")]), root_freqs=NUC_FREQS,
                        logl=res["logl"], site_index=np.arange(sites), site_logl=res["site_logl"])
    print(f"{name}: logL = {res['logl']!r}  sites stored {sites}")


def fullsize_partitions(name="full_gtr_32x50k_x8", taxa=32, sites=50000, n_part=8, seed0=5, threads=8):
    """configs[4]: GARD-style multi-partition GTR — 8 partitions of 32 taxa x 50 000 sites, each with its own tree and
    alignment, ONE likelihood function over all of them (syntax precedent: res/TemplateBatchFiles/REL/MultiplePartitions.bf
    builds `LikelihoodFunction lf = (filter_1, tree_1, filter_2, tree_2, ...)`).  Stored: the total log L, every
    partition's own log L (single-partition functions) and sampled per-site values of partition 0."""
    import tempfile
    tmp = tempfile.mkdtemp(prefix="hygold_")
    outp = os.path.join(tmp, "out.txt")
    rev = dict(AC=0.5, AT=0.4, CG=0.4, CT=1.2, GT=0.4)
    L = ["VERBOSITY_LEVEL = -1;", "PRINT_DIGITS = 17;"] + [f"global {k} = {v!r};" for k, v in rev.items()]
    L.append(hbl.nuc_model_block(NUC_FREQS))
    L.append("UseModel (NM);")
    crcs, part_ll = [], []
    pairs = []
    for k in range(n_part):
        syn = data.evolve(taxa, sites, 1, seed=seed0 + k, p_change=0.25)
        crcs.append(_crc(syn.states.astype(np.int16)))
        fasta = os.path.join(tmp, f"p{k}.fasta")
        hbl.write_fasta(fasta, syn.flat.leaf_names, syn.seqs)
        L.append(f"Tree t{k} = {tree.to_newick(syn.tree)};")
        L.append(f'DataSet ds{k} = ReadDataFile ("{fasta}");')
        L.append(f"DataSetFilter f{k} = CreateFilter (ds{k},1);")
        for n in syn.flat.branch_names():
            L.append(f"t{k}.{n}.t = 0.05;")
        pairs.append(f"f{k}, t{k}")
        if k == 0:
            site0 = syn
    L.append(f"LikelihoodFunction lf = ({', '.join(pairs)});")
    L.append(f"NUMBER_THREADS = {threads};")
    L.append("LFCompute (lf, LF_START_COMPUTE); LFCompute (lf, res0); LFCompute (lf, LF_DONE_COMPUTE);")
    L.append(f'fprintf ("{outp}", CLEAR_FILE, "LOGL ", Format (res0, 30, 17), "\n");')
    for k in range(n_part):
        L.append(f"LikelihoodFunction lf{k} = (f{k}, t{k}); LFCompute (lf{k}, LF_START_COMPUTE); LFCompute (lf{k}, r{k}); LFCompute (lf{k}, LF_DONE_COMPUTE);")
        L.append(f'fprintf ("{outp}", "PART ", Format (r{k}, 30, 17), "\n");')
    L.append("ConstructCategoryMatrix (sl_, lf0, SITE_LOG_LIKELIHOODS);")
    L.append(f'fprintf ("{outp}", "SITES ", Columns (sl_), "\n");')
    L.append(f'for (k_ = 0; k_ < Columns (sl_); k_ += 1) {{ fprintf ("{outp}", Format (sl_[k_], 30, 17), "\n"); }}')
    hbl.run_script("\n".join(L) + "\n", tmp, cpus=threads, timeout=3600.0)
    res = hbl.parse_output(outp)
    with open(outp) as fh:
        part_ll = [float(ln.split()[1]) for ln in fh if ln.startswith("PART ")]
    idx = _site_sample(sites, seed0)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="nuc_partitions", taxa=taxa, sites=sites, n_part=n_part,
                        seed0=seed0, p_change=0.25, states_crc=np.array(crcs), t=0.05,
                        rev=np.array([rev[k] for k in ("AC", "AT", "CG", "CT", "GT")]), root_freqs=NUC_FREQS,
                        logl=res["logl"], part_logl=np.array(part_ll), site_index=idx, site_logl_part0=res["site_logl"][idx])
    print(f"{name}: logL = {res['logl']!r}  sum of partitions = {sum(part_ll)!r}")


def fullsize_sweep(name="full_mg94_64x10k_sweep", taxa=64, codons=10000, seed=3, n=40, threads=16):
    """The parameter points bench.py's timed loop visits on the headline workload (omega = 0.3 + 0.001 k, k = 1 .. n, every
    branch at t = 0.05): the reference's log L at each of them, so that the TIMED path (hyphy_hip_build_q +
    hyphy_hip_evaluate_built on the tuned / forced schedules) is held to the reference under `pytest -m gpu`, not only inside
    bench.py's own parity block."""
    syn = data.evolve(taxa, codons, 3, seed=seed, p_change=0.04)
    flat = syn.flat
    bt = {nm: 0.05 for nm in flat.branch_names()}
    tmpl = models.mg94rev_template(POS_FREQS)
    pi = models.f3x4_codon_freqs(POS_FREQS)
    g = dict(R=0.3, **REV)
    res = hbl.evaluate(names=flat.leaf_names, seqs=syn.seqs, newick=tree.to_newick(syn.tree), unit=3,
                       model_block=hbl.codon_model_block(tmpl, pi), model_name="MGM", globals_=g, branch_t=bt,
                       sweep=dict(param="R", start=0.3, step=0.001, n=n, record=n), threads=threads, per_site=False, timeout=3600.0)
    sv = np.asarray(res["sweep_values"])
    assert len(sv) == n
    np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="codon_sweep", taxa=taxa, sites=codons, seed=seed, p_change=0.04,
                        states_crc=_crc(syn.states.astype(np.int16)), t=0.05, omega0=0.3, omega_step=0.001,
                        rev=np.array([REV[k] for k in ("AC", "AT", "CG", "CT", "GT")]), pos_freqs=POS_FREQS,
                        logl0=res["logl"], sweep_logl=sv)
    print(f"{name}: logL(0.3) = {res['logl']!r}, {n} sweep points, last = {sv[-1]!r}")


def fullsize_cases():
    fullsize_codon("full_mg94_32x5k", 32, 5000, seed=2)            # configs[1]
    fullsize_codon("full_mg94_64x10k", 64, 10000, seed=3)          # the headline metric's workload
    fullsize_codon("full_busted3_64x10k", 64, 10000, seed=3,       # configs[2]
                   classes=dict(weights=[0.7, 0.25, 0.05], values=[0.1 / 0.3, 1.0 / 0.3, 5.0 / 0.3]))
    fullsize_codon("full_mg94_128x100k", 128, 100000, seed=4)      # configs[3] (all 100 000 codons on one device in the test)
    fullsize_partitions()                                          # configs[4]
    fullsize_nuc()                                                 # configs[0]
    fullsize_mix3()                                                # configs[2] as an explicit-form branch-site mixture


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "mixture":
        os.makedirs(OUT, exist_ok=True)
        mixture_case()
        mixture_case("codon_mix3", 20, 80, seed=43, omegas=(0.05, 0.8, 6.0), weights=(0.6, 0.3, 0.1))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fel":
        if not hbl.have_reference():
            raise SystemExit("oracle/_ref/hyphy missing: run `make -f oracle/Makefile.ref -j8` first")
        os.makedirs(OUT, exist_ok=True)
        fel_case()
        fel_case("ref_fel_10x48_all", n_taxa=10, n_codons=48, seed=203, branches="All")   # (every branch tested: no nuisance rate)
        fel_case("ref_meme_12x60", analysis="MEME")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fubar":
        if not hbl.have_reference():
            raise SystemExit("oracle/_ref/hyphy missing: run `make -f oracle/Makefile.ref -j8` first")
        os.makedirs(OUT, exist_ok=True)
        fubar_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "busted":
        if not hbl.have_reference():
            raise SystemExit("oracle/_ref/hyphy missing: run `make -f oracle/Makefile.ref -j8` first")
        os.makedirs(OUT, exist_ok=True)
        busted_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        if not hbl.have_reference():
            raise SystemExit("oracle/_ref/hyphy missing: run `make -f oracle/Makefile.ref -j8` first")
        os.makedirs(OUT, exist_ok=True)
        fullsize_sweep()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize_hky":
        fullsize_nuc()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize_mix3":
        if not hbl.have_reference():
            raise SystemExit("oracle/_ref/hyphy missing: run `make -f oracle/Makefile.ref -j8` first")
        os.makedirs(OUT, exist_ok=True)
        fullsize_mix3()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize":
        if not hbl.have_reference():
            raise SystemExit("oracle/_ref/hyphy missing: run `make -f oracle/Makefile.ref -j8` first")
        os.makedirs(OUT, exist_ok=True)
        fullsize_cases()
        return
    if not hbl.have_reference():
        raise SystemExit("oracle/_ref/hyphy missing: run `make -f oracle/Makefile.ref -j8` first")
    os.makedirs(OUT, exist_ok=True)
    codon_case("codon_small", 8, 40, seed=11)
    codon_case("codon_ambig", 8, 60, seed=12, missing=0.05)
    codon_case("codon_deep", 120, 12, seed=13, ladder=True, tlo=0.2, thi=0.6, p_change=0.3)
    codon_case("codon_cat3", 10, 50, seed=14,
               category=dict(name="rc", weights=[0.7, 0.25, 0.05], values=[0.1, 1.0, 5.0]))
    codon_case("codon_wide", 64, 60, seed=15)        # random 64-taxon tree: deep nesting, many pending subtrees
    nuc_case("nuc_wide", 100, 200, seed=24)
    nuc_case("nuc_small", 8, 300, seed=21)
    nuc_case("nuc_ambig", 12, 200, seed=22, missing=0.05, rev=dict(AC=0.5, AT=0.4, CG=0.4, CT=1.2, GT=0.4))
    nuc_case("nuc_deep", 300, 40, seed=23, ladder=True, tlo=0.1, thi=0.5, p_change=0.25)
    expm_cases()
    marginal_support_case()
    mixture_case()
    mixture_case("codon_mix3", 20, 80, seed=43, omegas=(0.05, 0.8, 6.0), weights=(0.6, 0.3, 0.1))
    reference_test_case()
    reference_test_case_nuc()


if __name__ == "__main__":
    main()
