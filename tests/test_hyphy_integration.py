"""End-to-end drop-in check on the GPU: the REAL HyPhy host (reference sources + the adapter of
INTEGRATION.md, built by integration/build.py into integration/_build/hyphy_hip) runs HBL batch
files with its ComputeBlock routed through libhyphy_hip.so.  Compared with the golden vectors of the
unmodified reference and — for a full Optimize() fit — with the unmodified reference binary run
side by side on the box.  Both binaries are prebuilt in the build container (they travel with the
snapshot); nothing here reads /root/reference."""
import os
import re

import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_BIN = os.path.join(ROOT, "integration", "_build", "hyphy_hip")
ENV = {"HYPHY_HIP": "1", "HYPHY_HIP_VERBOSE": "1"}


def _need_binaries():
    from oracle import hbl
    if not (os.path.isfile(HIP_BIN) and hbl.have_reference()):
        pytest.skip("integration/_build/hyphy_hip or oracle/_ref/hyphy not built (build container: "
                    "python integration/build.py)")


def _case(kind, n_taxa, n_sites, seed, category=None, ladder=False, **kw):
    """Same deterministic inputs as oracle/make_golden.py."""
    from hyphy_amd import data, models, tree
    from oracle import hbl, make_golden as mg
    tr = tree.caterpillar_tree(n_taxa) if ladder else None
    syn = data.evolve(n_taxa, n_sites, 3 if kind == "codon" else 1, seed=seed, tree=tr, **kw)
    flat = syn.flat
    if kind == "codon":
        bt = mg.branch_lengths(flat, seed + 7, 0.02, 0.12)
        rate = "t" if category is None else f"{category['name']}*t"
        block = hbl.codon_model_block(models.mg94rev_template(mg.POS_FREQS), models.f3x4_codon_freqs(mg.POS_FREQS),
                                      rate_expr=rate)
        args = dict(unit=3, model_block=block, model_name="MGM", globals_=dict(R=0.3, **mg.REV))
    else:
        bt = mg.branch_lengths(flat, seed + 7, 0.02, 0.2)
        args = dict(unit=1, model_block=hbl.nuc_model_block(mg.NUC_FREQS), model_name="NM",
                    globals_=dict(models.hky85_rev(0.35)))
    return dict(names=flat.leaf_names, seqs=syn.seqs, newick=tree.to_newick(syn.tree), branch_t=bt,
                category=category, **args)


def _cached_calls(stdout):
    m = re.findall(r"\(\+ (\d+) through the branch cache\)", stdout)
    return max(int(x) for x in m) if m else 0


def _device_calls(stdout):
    m = re.findall(r"\[hyphy_hip\] (\d+) ComputeBlock evaluations ran on the device", stdout)
    return max(int(x) for x in m) if m else 0


@pytest.mark.parametrize("name,kind,taxa,sites,seed", [("codon_small", "codon", 8, 40, 11),
                                                       ("codon_wide", "codon", 64, 60, 15),
                                                       ("nuc_small", "nuc", 8, 300, 21)])
def test_hbl_lfcompute_through_device_matches_reference_goldens(name, kind, taxa, sites, seed):
    _need_binaries()
    from oracle import hbl
    fx = common.load(name)
    res = hbl.evaluate(binary=HIP_BIN, extra_env=ENV, **_case(kind, taxa, sites, seed))
    assert _device_calls(res["stdout"]) > 0, "the device path was not taken"
    ref = float(fx["logl"])
    assert abs(res["logl"] - ref) <= 1e-10 * abs(ref)
    assert np.max(np.abs(res["site_logl"] - fx["site_logl"]) / np.abs(fx["site_logl"])) < 1e-10


def test_hbl_hky_8x1k_at_stated_size_through_device():
    """configs[0] at its stated size (HKY85, 8 taxa x 1 000 sites, bench.py's `hky_8x1k` alignment) through the real host:
    LFCompute and the per-site values against the unmodified reference's (tests/golden/full_hky_8x1k.npz)."""
    _need_binaries()
    from hyphy_amd import data, models, tree
    from oracle import hbl, make_golden as mg
    fx = common.load("full_hky_8x1k")
    syn = data.evolve(int(fx["taxa"]), int(fx["sites"]), 1, seed=int(fx["seed"]), p_change=float(fx["p_change"]))
    bt = {n: float(fx["t"]) for n in syn.flat.branch_names()}
    res = hbl.evaluate(binary=HIP_BIN, extra_env=ENV, names=syn.flat.leaf_names, seqs=syn.seqs, newick=tree.to_newick(syn.tree),
                       unit=1, model_block=hbl.nuc_model_block(mg.NUC_FREQS), model_name="NM",
                       globals_=dict(models.hky85_rev(float(fx["kappa"]))), branch_t=bt)
    assert _device_calls(res["stdout"]) > 0, "the device path was not taken"
    ref = float(fx["logl"])
    assert abs(res["logl"] - ref) <= 1e-10 * abs(ref)
    assert np.max(np.abs(res["site_logl"] - fx["site_logl"]) / np.abs(fx["site_logl"])) < 1e-10


def test_hbl_rate_categories_through_device():
    """3 discrete rate classes: HyPhy's own PopulateConditionalProbabilities / SumUpSiteLikelihoods mix the
    per-class (l_s, c_s) pairs that the device returns through siteRes / siteCorrections."""
    _need_binaries()
    from oracle import hbl
    fx = common.load("codon_cat3")
    cat = dict(name="rc", weights=[0.7, 0.25, 0.05], values=[0.1, 1.0, 5.0])
    res = hbl.evaluate(binary=HIP_BIN, extra_env=ENV, **_case("codon", 10, 50, 14, category=cat))
    assert _device_calls(res["stdout"]) > 0
    ref = float(fx["logl"])
    assert abs(res["logl"] - ref) <= 1e-10 * abs(ref)
    assert np.max(np.abs(res["site_logl"] - fx["site_logl"]) / np.abs(fx["site_logl"])) < 1e-10


def test_hbl_deep_tree_rescaling_through_device():
    _need_binaries()
    from oracle import hbl, make_golden as mg
    from hyphy_amd import data, models, tree
    fx = common.load("codon_deep")
    syn = data.evolve(120, 12, 3, seed=13, p_change=0.3, tree=tree.caterpillar_tree(120))
    bt = mg.branch_lengths(syn.flat, 20, 0.2, 0.6)
    res = hbl.evaluate(binary=HIP_BIN, extra_env=ENV, names=syn.flat.leaf_names, seqs=syn.seqs,
                       newick=tree.to_newick(syn.tree), unit=3,
                       model_block=hbl.codon_model_block(models.mg94rev_template(mg.POS_FREQS),
                                                         models.f3x4_codon_freqs(mg.POS_FREQS)),
                       model_name="MGM", globals_=dict(R=0.3, **mg.REV), branch_t=bt)
    assert _device_calls(res["stdout"]) > 0
    ref = float(fx["logl"])
    assert abs(res["logl"] - ref) <= 1e-10 * abs(ref)


def test_hbl_optimize_through_device_matches_cpu_fit():
    """A complete maximum-likelihood fit (HBL Optimize: hundreds of ComputeBlock calls, most of them
    partial updates driven by DetermineNodesForUpdate) reaches the same optimum as the CPU reference."""
    _need_binaries()
    from oracle import hbl
    case = _case("codon", 8, 40, 11)
    cpu = hbl.evaluate(optimize=True, **case)
    gpu = hbl.evaluate(optimize=True, binary=HIP_BIN, extra_env=ENV, **case)
    assert _device_calls(gpu["stdout"]) > 50
    # the optimiser's one-branch line searches went through the device branch cache, driven by
    # the reference's own computedLocalUpdatePolicy state machine
    assert _cached_calls(gpu["stdout"]) > 20, gpu["stdout"][-600:]
    assert abs(gpu["opt_logl"] - cpu["opt_logl"]) <= 2e-3       # 2 x OPTIMIZATION_PRECISION, the reference's own test bar
    assert abs(gpu["logl"] - cpu["logl"]) <= 1e-10 * abs(cpu["logl"])


@pytest.mark.parametrize("mode", ["joint", "marginal"])
def test_hbl_ancestral_reconstruction_through_adapter_matches_cpu(mode):
    """ReconstructAncestors after device evaluations.  Joint reconstruction recomputes its own tables on the host;
    MARGINAL pins node states (ComputeBlock with branchIndex >= 0, likefunc2.cpp:932-1040): routed to
    hyphy_hip_set_pinned_states + hyphy_hip_evaluate (a CPU fallback would have to recompute every node, because the
    host caches are never filled while the device path is active — the adapter does that too)."""
    _need_binaries()
    import tempfile
    from oracle import hbl
    case = _case("codon", 8, 40, 11)

    def run(binary, env):
        tmp = tempfile.mkdtemp(prefix="anc_")
        fasta, outp, ancp = (os.path.join(tmp, n) for n in ("aln.fasta", "out.txt", "anc.txt"))
        hbl.write_fasta(fasta, case["names"], case["seqs"])
        txt = hbl.build_script(fasta=fasta, newick=case["newick"], unit=case["unit"], model_block=case["model_block"],
                               model_name=case["model_name"], globals_=case["globals_"], branch_t=case["branch_t"],
                               out_path=outp, per_site=False)
        txt += ("DataSet anc = ReconstructAncestors (lf" + (", MARGINAL" if mode == "marginal" else "") + ");\n"
                "DataSetFilter af = CreateFilter (anc, 1);\nDATA_FILE_PRINT_FORMAT = 9;\n"
                f'fprintf ("{ancp}", CLEAR_FILE, af);\n')
        out = hbl.run_script(txt, tmp, binary=binary, extra_env=env)
        return open(ancp).read(), out

    cpu, _ = run(None, None)
    gpu, stdout = run(HIP_BIN, ENV)
    # MARGINAL: one pinned device evaluation per internal node and state (6 x 60 here) on top of the baseline
    assert _device_calls(stdout) > (300 if mode == "marginal" else 0)
    assert len(cpu) > 100 and cpu == gpu


def _deferred(stdout):
    m = re.findall(r"(\d+) matrix exponentials moved to the device", stdout)
    return max(int(x) for x in m) if m else 0


def test_hbl_optimize_exponentiates_on_the_device_and_leaves_host_matrices_current():
    """Mode B: while Optimize runs, ExponentiateMatrices hands its queue of rate matrices to the
    adapter (device expm) instead of exponentiating on the host; when Optimize returns the host-side transition
    matrices are brought up to date, so a consumer that reads them directly — joint ancestral reconstruction inside
    the same LF_START_COMPUTE bracket — sees the fitted model.  Compared with the same binary in mode A
    (HYPHY_HIP_DEVICE_EXPM=0: host exponentials) and with the unmodified reference."""
    _need_binaries()
    import tempfile
    from oracle import hbl
    case = _case("codon", 8, 40, 11)

    def run(binary, env):
        tmp = tempfile.mkdtemp(prefix="modeb_")
        fasta, outp, ancp = (os.path.join(tmp, n) for n in ("aln.fasta", "out.txt", "anc.txt"))
        hbl.write_fasta(fasta, case["names"], case["seqs"])
        txt = hbl.build_script(fasta=fasta, newick=case["newick"], unit=case["unit"], model_block=case["model_block"],
                               model_name=case["model_name"], globals_=case["globals_"], branch_t=case["branch_t"],
                               out_path=outp, per_site=False)
        tail = ("OPTIMIZATION_PRECISION = 0.001; VERBOSITY_LEVEL = -1;\nOptimize (m2_, lf);\n"
                f'fprintf ("{outp}", "OPT_LOGL ", Format (m2_[1][0], 30, 17), "\\n");\n'
                "DataSet anc = ReconstructAncestors (lf);\nDataSetFilter af = CreateFilter (anc, 1);\nDATA_FILE_PRINT_FORMAT = 9;\n"
                f'fprintf ("{ancp}", CLEAR_FILE, af);\n'
                "LFCompute (lf, LF_DONE_COMPUTE);\n")
        assert txt.count("LFCompute (lf, LF_DONE_COMPUTE);\n") == 1
        txt = txt.replace("LFCompute (lf, LF_DONE_COMPUTE);\n", tail)
        out = hbl.run_script(txt, tmp, binary=binary, extra_env=env)
        return open(ancp).read(), hbl.parse_output(outp), out

    anc_cpu, res_cpu, _ = run(None, None)
    anc_a, res_a, out_a = run(HIP_BIN, dict(ENV, HYPHY_HIP_DEVICE_EXPM="0"))
    anc_b, res_b, out_b = run(HIP_BIN, ENV)
    assert _deferred(out_a) == 0
    assert _deferred(out_b) > 100, out_b[-400:]
    assert _device_calls(out_b) > 50
    assert abs(res_b["opt_logl"] - res_cpu["opt_logl"]) <= 2e-3 and abs(res_a["opt_logl"] - res_cpu["opt_logl"]) <= 2e-3
    assert len(anc_cpu) > 100 and anc_b == anc_a == anc_cpu


def test_hbl_optimize_with_rate_categories_through_device():
    """Optimize of a model with 3 rate classes: ComputeBlock runs once per class (catID >= 0), so mode B stashes the
    rate matrices per class; the fit must land on the CPU optimum."""
    _need_binaries()
    from oracle import hbl
    cat = dict(name="rc", weights=[0.7, 0.25, 0.05], values=[0.1, 1.0, 5.0])
    case = _case("codon", 8, 40, 11, category=cat)
    cpu = hbl.evaluate(optimize=True, per_site=False, **case)
    gpu = hbl.evaluate(optimize=True, per_site=False, binary=HIP_BIN, extra_env=ENV, **case)
    assert _device_calls(gpu["stdout"]) > 50
    assert _deferred(gpu["stdout"]) > 100
    assert abs(gpu["opt_logl"] - cpu["opt_logl"]) <= 2e-3
    assert abs(gpu["logl"] - cpu["logl"]) <= 1e-10 * abs(cpu["logl"])


def _cat_batches(stdout):
    m = re.findall(r"rate classes: (\d+) category loops answered by ONE device evaluation of all classes, (\d+) classes evaluated one by one", stdout)
    return (max(int(x[0]) for x in m), max(int(x[1]) for x in m)) if m else (0, 0)


@pytest.mark.parametrize("expm", ["always", "0"])
def test_hbl_rate_class_loop_answered_by_one_device_evaluation(expm):
    """r06: the host's category loop (PopulateConditionalProbabilities, weighted-sum mode, likefunc2.cpp:484-908) calls ComputeBlock
    once per rate class; the adapter collects the classes and answers the loop with ONE hyphy_hip_evaluate_categories(_built_sites)
    call (template rows with device exponentials — expm = always — or the host's own transition matrices — 0).  The benchmark's
    sweep (every matrix of every class changes at every point) against the unmodified binary at every point, and against the
    adapter with the batching off (HYPHY_HIP_CAT_BATCH=0: one device evaluation per class, mixed by the host)."""
    _need_binaries()
    from oracle import hbl
    cat = dict(name="rc", weights=[0.7, 0.25, 0.05], values=[0.1, 1.0, 5.0])
    case = _case("codon", 12, 70, 35, category=cat)
    sweep = dict(param="R", start=0.3, step=0.01, n=20, record=20)
    env = dict(ENV, HYPHY_HIP_DEVICE_EXPM=expm)
    cpu = hbl.evaluate(sweep=sweep, per_site=True, **case)
    gpu = hbl.evaluate(sweep=sweep, per_site=True, binary=HIP_BIN, extra_env=env, **case)
    off = hbl.evaluate(sweep=sweep, per_site=True, binary=HIP_BIN, extra_env=dict(env, HYPHY_HIP_CAT_BATCH="0"), **case)
    nb, ns = _cat_batches(gpu["stdout"])
    assert nb >= 15 and ns == 0, gpu["stdout"][-800:]
    assert _cat_batches(off["stdout"]) == (0, 0)
    assert np.max(np.abs(gpu["sweep_values"] - cpu["sweep_values"]) / np.abs(cpu["sweep_values"])) < 1e-10
    assert np.max(np.abs(off["sweep_values"] - cpu["sweep_values"]) / np.abs(cpu["sweep_values"])) < 1e-10
    assert abs(gpu["logl"] - cpu["logl"]) <= 1e-10 * abs(cpu["logl"])
    # per-site values come through the same hook with the mixed per-pattern values (ConstructCategoryMatrix does not ask for the sum)
    assert np.max(np.abs(gpu["site_logl"] - cpu["site_logl"]) / np.abs(cpu["site_logl"])) < 1e-10


def test_hbl_optimize_with_rate_classes_batched_lands_on_the_cpu_optimum():
    """Optimize of a 3-class model with the class loop batched: full passes go through one device evaluation of all classes, the
    one-branch line searches through the branch cache (those calls are never collected) — same optimum as the unmodified binary."""
    _need_binaries()
    from oracle import hbl
    cat = dict(name="rc", weights=[0.6, 0.3, 0.1], values=[0.2, 1.0, 4.0])
    case = _case("codon", 10, 50, 17, category=cat)
    cpu = hbl.evaluate(optimize=True, per_site=False, **case)
    gpu = hbl.evaluate(optimize=True, per_site=False, binary=HIP_BIN, extra_env=ENV, **case)
    nb, _ = _cat_batches(gpu["stdout"])
    assert nb > 10, gpu["stdout"][-800:]
    assert abs(gpu["opt_logl"] - cpu["opt_logl"]) <= 2e-3
    assert abs(gpu["logl"] - cpu["logl"]) <= 1e-10 * abs(cpu["logl"])


def test_reference_known_answer_smallcodon_through_device():
    """The reference's own known-answer test SimpleOptimizations/SmallCodon.bf (HIV-1 RT, 8 x 440 codons, MG94x012232,
    expected maximised log L = -3189.516375) fitted through the adapter: every ComputeBlock on the device, exponentials
    on the device, line searches through the device branch cache.  Tolerance = the reference harness's own
    (2 x OPTIMIZATION_PRECISION)."""
    _need_binaries()
    from oracle import hbl
    from hyphy_amd import models, tree
    fx = common.load("ref_smallcodon")
    flat = tree.flatten(tree.parse_newick(str(fx["newick"]) + ";"))
    res = hbl.evaluate(binary=HIP_BIN, extra_env=ENV, names=[str(x) for x in fx["names"]], seqs=[str(x) for x in fx["seqs"]],
                       newick=str(fx["newick"]), unit=3,
                       model_block=hbl.codon_model_block(models.mg94rev_template(fx["pos_freqs"]), fx["root_freqs"]),
                       model_name="MGM", globals_=dict(R=1.0, AC=1.0, AT=1.0, CG=1.0, CT=1.0, GT=1.0),
                       branch_t={n: 0.1 for n in flat.branch_names()}, optimize=True, per_site=False,
                       constraints=dict(CG="AT", GT="AT"))
    assert _device_calls(res["stdout"]) > 50 and _deferred(res["stdout"]) > 100
    assert abs(res["opt_logl"] - float(fx["expected_opt_logl"])) <= 2e-3
    assert abs(res["logl"] - float(fx["logl"])) <= 1e-10 * abs(float(fx["logl"]))


def _template_evals(stdout):
    m = re.findall(r"template mode: (\d+) evaluations took their rate matrices as coefficients \(K = (\d+)\), (\d+) RecomputeMatrix calls skipped", stdout)
    return (max(int(x[0]) for x in m), int(m[-1][1]), max(int(x[2]) for x in m)) if m else (0, 0, 0)


def test_hbl_optimize_in_template_mode():
    """Template mode of the adapter (SURVEY 8f-3 through the real host): while Optimize runs, the host evaluates the rate
    matrix of K + 1 branches per ExponentiateMatrices call (K = 1 local parameter: t), the adapter derives the templates of
    that evaluation from them, verifies on one more branch and sends every other branch as K coefficients
    (hyphy_hip_build_q).  Same optimum as the CPU reference and as the adapter with HYPHY_HIP_TEMPLATES=0."""
    _need_binaries()
    from oracle import hbl
    case = _case("codon", 16, 60, 31)
    cpu = hbl.evaluate(optimize=True, **case)
    gpu = hbl.evaluate(optimize=True, binary=HIP_BIN, extra_env=ENV, **case)
    off = hbl.evaluate(optimize=True, binary=HIP_BIN, extra_env=dict(ENV, HYPHY_HIP_TEMPLATES="0"), **case)
    n_eval, K, n_skip = _template_evals(gpu["stdout"])
    assert K == 1 and n_eval > 10 and n_skip > 10 * n_eval, gpu["stdout"][-800:]
    assert _template_evals(off["stdout"])[0] == 0
    assert abs(gpu["opt_logl"] - cpu["opt_logl"]) <= 2e-3
    assert abs(off["opt_logl"] - cpu["opt_logl"]) <= 2e-3
    assert abs(gpu["logl"] - cpu["logl"]) <= 1e-10 * abs(cpu["logl"])


def test_hbl_lfcompute_sweep_in_template_mode_matches_reference_at_every_point():
    """The benchmark's own loop through the real host: R swept, every branch's matrix changes at every point, mode B
    forced (HYPHY_HIP_DEVICE_EXPM=always) -> template mode after the first evaluation.  Every value of the sweep against
    the unmodified binary."""
    _need_binaries()
    from oracle import hbl
    case = _case("codon", 24, 80, 33)
    sweep = dict(param="R", start=0.3, step=0.01, n=25, record=25)
    cpu = hbl.evaluate(sweep=sweep, per_site=False, **case)
    gpu = hbl.evaluate(sweep=sweep, per_site=False, binary=HIP_BIN, extra_env=dict(ENV, HYPHY_HIP_DEVICE_EXPM="always"), **case)
    n_eval, K, _ = _template_evals(gpu["stdout"])
    assert K == 1 and n_eval >= 20, gpu["stdout"][-800:]
    assert np.max(np.abs(gpu["sweep_values"] - cpu["sweep_values"]) / np.abs(cpu["sweep_values"])) < 1e-10


def test_hbl_two_partitions_on_the_device():
    """A likelihood function with two partitions (own alignment and tree each; REL/MultiplePartitions.bf syntax): both go
    to the device (partition i -> device i mod n), Compute() enqueues both before it collects the first; LFCompute value,
    per-site values and a complete Optimize against the unmodified binary."""
    _need_binaries()
    from oracle import hbl
    a, b = _case("codon", 8, 40, 11), _case("codon", 10, 30, 12)
    xp = [dict(names=b["names"], seqs=b["seqs"], newick=b["newick"], branch_t=b["branch_t"])]
    cpu = hbl.evaluate(extra_partitions=xp, **a)                       # value and per-site values at the start point
    gpu = hbl.evaluate(extra_partitions=xp, binary=HIP_BIN, extra_env=ENV, **a)
    assert "partition 0 of 2 -> device" in gpu["stdout"] and "partition 1 of 2 -> device" in gpu["stdout"], gpu["stdout"][-800:]
    assert abs(gpu["logl"] - cpu["logl"]) <= 1e-10 * abs(cpu["logl"])
    assert np.max(np.abs(gpu["site_logl"] - cpu["site_logl"]) / np.abs(cpu["site_logl"])) < 1e-10
    cpu = hbl.evaluate(extra_partitions=xp, optimize=True, per_site=False, **a)   # a complete fit
    gpu = hbl.evaluate(extra_partitions=xp, optimize=True, per_site=False, binary=HIP_BIN, extra_env=ENV, **a)
    assert _device_calls(gpu["stdout"]) > 50
    assert abs(gpu["opt_logl"] - cpu["opt_logl"]) <= 2e-3


def _mixture_case(n_taxa=12, n_codons=60, seed=41):
    from hyphy_amd import data, models, tree
    from oracle import hbl, make_golden as mg
    syn = data.evolve(n_taxa, n_codons, 3, seed=seed)
    bt = mg.branch_lengths(syn.flat, seed + 7, 0.02, 0.3)
    block = hbl.codon_mixture_model_block(models.mg94rev_template(mg.POS_FREQS), models.f3x4_codon_freqs(mg.POS_FREQS),
                                          ["R1", "R2"], ["W1", "(1-W1)"])
    return dict(names=syn.flat.leaf_names, seqs=syn.seqs, newick=tree.to_newick(syn.tree), unit=3, model_block=block,
                model_name="MGM", globals_=dict(R1=0.1, R2=2.5, W1=0.7, **mg.REV), branch_t=bt, upper_bounds=dict(W1=1.0))


def test_hbl_explicit_form_mixture_through_device():
    """BUSTED / BS-REL shaped model in the reference's explicit form (`Model = ("Exp(Q1)*W1+Exp(Q2)*(1-W1)", freqs,
    EXPLICIT_FORM_MATRIX_EXPONENTIAL)`): LFCompute value and per-site values against the reference's golden, then a
    complete Optimize in which the adapter takes the queued Exp() arguments (mode B, mixture mode): exponentials and
    mixing on the device, the weights from the model formula itself; same optimum as the unmodified binary."""
    _need_binaries()
    from oracle import hbl
    fx = common.load("codon_mix2")
    case = _mixture_case()
    res = hbl.evaluate(binary=HIP_BIN, extra_env=ENV, **case)
    assert _device_calls(res["stdout"]) > 0
    ref = float(fx["logl"])
    assert abs(res["logl"] - ref) <= 1e-10 * abs(ref)
    assert np.max(np.abs(res["site_logl"] - fx["site_logl"]) / np.abs(fx["site_logl"])) < 1e-10
    cpu = hbl.evaluate(optimize=True, per_site=False, **case)
    gpu = hbl.evaluate(optimize=True, per_site=False, binary=HIP_BIN, extra_env=ENV, **case)
    m = re.findall(r"mixture mode: (\d+) evaluations exponentiated and mixed their (\d+)-component", gpu["stdout"])
    mt = re.findall(r"mixture template mode: (\d+) evaluations took their component rate matrices as coefficients \(M = (\d+), K = (\d+)\), (\d+) RecomputeMatrix", gpu["stdout"])
    n_dense, n_rows = (int(m[-1][0]) if m else 0), (int(mt[-1][0]) if mt else 0)
    assert n_dense + n_rows > 10 and (not m or int(m[-1][1]) == 2), gpu["stdout"][-1200:]
    # (the reference's own tolerance for optimised values is 2 x OPTIMIZATION_PRECISION = 2e-3; the two optimisers see values that
    #  differ in the 10th digit and may stop a step apart: the device's optimum must not be WORSE than the reference's by more than
    #  that, and within 1e-2 of it)
    assert gpu["opt_logl"] >= cpu["opt_logl"] - 2e-3 and abs(gpu["opt_logl"] - cpu["opt_logl"]) <= 1e-2, (gpu["opt_logl"], cpu["opt_logl"])
    # sweeps with mode B forced, every point against the unmodified binary: the mixture weight, then a component's omega (a global
    # inside the component rate matrices: the component templates must follow it).  r04: after two dense calls the components
    # reach the device as rows of branch-local parameters over per-call templates (mixture template mode).
    for param, start in (("W1", 0.3), ("R1", 0.05)):
        sweep = dict(param=param, start=start, step=0.02, n=20, record=20)
        cpu = hbl.evaluate(sweep=sweep, per_site=False, **case)
        gpu = hbl.evaluate(sweep=sweep, per_site=False, binary=HIP_BIN, extra_env=dict(ENV, HYPHY_HIP_DEVICE_EXPM="always"), **case)
        assert np.max(np.abs(gpu["sweep_values"] - cpu["sweep_values"]) / np.abs(cpu["sweep_values"])) < 1e-10
        mt = re.findall(r"mixture template mode: (\d+) evaluations took their component rate matrices as coefficients \(M = (\d+), K = (\d+)\), (\d+) RecomputeMatrix", gpu["stdout"])
        if param == "R1":   # (a weight sweep dirties no rate matrix: nothing to build)
            assert mt and int(mt[-1][0]) >= 10 and int(mt[-1][1]) == 2 and int(mt[-1][2]) == 1 and int(mt[-1][3]) > 0, gpu["stdout"][-1500:]
    # and with the rows switched off: dense component matrices (r02's mixture mode), same numbers
    gpu = hbl.evaluate(sweep=sweep, per_site=False, binary=HIP_BIN, extra_env=dict(ENV, HYPHY_HIP_DEVICE_EXPM="always", HYPHY_HIP_TEMPLATES="0"), **case)
    assert "mixture template mode" not in gpu["stdout"]
    assert np.max(np.abs(gpu["sweep_values"] - cpu["sweep_values"]) / np.abs(cpu["sweep_values"])) < 1e-10


def test_hbl_explicit_form_mixture_at_stated_size_through_device():
    """configs[2] as BUSTED runs it, through the real host: 64 taxa x 10 000 codons, three omega classes in the reference's explicit
    form — LFCompute and a short sweep of a component's omega (the adapter's mixture template mode: the component rate matrices reach
    the device as coefficient rows), against the committed value of the unmodified reference (tests/golden/full_mix3_64x10k.npz)."""
    _need_binaries()
    from hyphy_amd import data, models, tree
    from oracle import hbl, make_golden as mg
    fx = common.load("full_mix3_64x10k")
    syn = data.evolve(int(fx["taxa"]), int(fx["sites"]), 3, seed=int(fx["seed"]), p_change=float(fx["p_change"]))
    om, w = fx["omegas"], fx["weights"]
    block = hbl.codon_mixture_model_block(models.mg94rev_template(mg.POS_FREQS), models.f3x4_codon_freqs(mg.POS_FREQS),
                                          ["R1", "R2", "R3"], ["W1", "W2", "(1-W1-W2)"])
    case = dict(names=syn.flat.leaf_names, seqs=syn.seqs, newick=tree.to_newick(syn.tree), unit=3, model_block=block, model_name="MGM",
                globals_=dict(R1=float(om[0]), R2=float(om[1]), R3=float(om[2]), W1=float(w[0]), W2=float(w[1]), **mg.REV),
                branch_t={n: float(fx["t"]) for n in syn.flat.branch_names()}, upper_bounds=dict(W1=1.0, W2=1.0))
    res = hbl.evaluate(binary=HIP_BIN, extra_env=dict(ENV, HYPHY_HIP_DEVICE_EXPM="always"), threads=1, **case)
    assert _device_calls(res["stdout"]) > 0
    ref = float(fx["logl"])
    assert abs(res["logl"] - ref) <= 1e-10 * abs(ref), (res["logl"], ref)
    got = res["site_logl"][fx["site_index"]]
    assert np.max(np.abs(got - fx["site_logl"]) / np.abs(fx["site_logl"])) < 1e-10
    # a sweep that dirties one component of every branch (mixture template mode after the adapter's learning calls), every point against
    # the unmodified binary run beside it
    sweep = dict(param="R2", start=float(om[1]), step=0.01, n=12, record=12)
    gpu = hbl.evaluate(sweep=sweep, per_site=False, binary=HIP_BIN, extra_env=dict(ENV, HYPHY_HIP_DEVICE_EXPM="always"), threads=1, **case)
    cpu = hbl.evaluate(sweep=sweep, per_site=False, threads=16, **case)
    assert np.max(np.abs(gpu["sweep_values"] - cpu["sweep_values"]) / np.abs(cpu["sweep_values"])) < 1e-10, (gpu["sweep_values"], cpu["sweep_values"])
    assert "mixture template mode" in gpu["stdout"], gpu["stdout"][-1500:]


def test_hbl_spmd_site_shard_one_rank_through_device():
    """SPMD site sharding of the adapter (one host process per GPU, every process runs the same batch file,
    HYPHY_HIP_WORLD / HYPHY_HIP_RANK): with a world of ONE — what a single-GPU box can run — the partition goes
    through hyphy_hip_comm_init_rank and every evaluation through hyphy_hip_evaluate*_allreduce; LFCompute, a sweep in mode B
    (template mode: hyphy_hip_evaluate_built_allreduce) and a complete Optimize must give the unmodified binary's numbers."""
    _need_binaries()
    from oracle import hbl
    fx = common.load("codon_wide")
    case = _case("codon", 64, 60, 15)
    env = dict(ENV, HYPHY_HIP_WORLD="1", HYPHY_HIP_RANK="0")
    res = hbl.evaluate(binary=HIP_BIN, extra_env=env, per_site=False, **case)
    assert "SPMD site shard" in res["stdout"], res["stdout"][-1500:]
    assert _device_calls(res["stdout"]) > 0
    ref = float(fx["logl"])
    assert abs(res["logl"] - ref) <= 1e-10 * abs(ref)
    sweep = dict(param="R", start=0.3, step=0.01, n=12, record=12)
    cpu = hbl.evaluate(sweep=sweep, per_site=False, **case)
    gpu = hbl.evaluate(sweep=sweep, per_site=False, binary=HIP_BIN, extra_env=dict(env, HYPHY_HIP_DEVICE_EXPM="always"), **case)
    assert np.max(np.abs(gpu["sweep_values"] - cpu["sweep_values"]) / np.abs(cpu["sweep_values"])) < 1e-10
    cpu = hbl.evaluate(optimize=True, per_site=False, **case)
    gpu = hbl.evaluate(optimize=True, per_site=False, binary=HIP_BIN, extra_env=env, **case)
    assert abs(gpu["opt_logl"] - cpu["opt_logl"]) <= 2e-3, (gpu["opt_logl"], cpu["opt_logl"])


def test_hbl_spmd_site_shard_two_ranks(tmp_path):
    """The same with two host processes on two GPUs (skipped on smaller boxes): both run the same batch file, each holds half
    of the patterns, the RCCL unique id travels through a file, both report the whole alignment's log-likelihood."""
    _need_binaries()
    from hyphy_amd import hip
    from oracle import hbl
    if hip.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import threading
    fx = common.load("codon_wide")
    case = _case("codon", 64, 60, 15)
    uid = str(tmp_path / "uid.bin")
    out = {}

    def run(rank):
        env = dict(ENV, HYPHY_HIP_WORLD="2", HYPHY_HIP_RANK=str(rank), HYPHY_HIP_UID_FILE=uid, HYPHY_HIP_RUN_ID=str(os.getpid()))
        # (sweep inside an LF_START_COMPUTE bracket, then Optimize: two SetupLFCaches, i.e. two communicators per process —
        #  each is a rendezvous of its own, "<uid file>.<n>")
        out[rank] = hbl.evaluate(binary=HIP_BIN, extra_env=env, per_site=False, optimize=True,
                                 sweep=dict(param="R", start=0.3, step=0.01, n=8, record=8), **case)

    th = [threading.Thread(target=run, args=(r,)) for r in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=900)
    ref = float(fx["logl"])
    for r in (0, 1):
        assert abs(out[r]["logl"] - ref) <= 1e-10 * abs(ref), (r, out[r]["logl"], ref)
    assert np.array_equal(out[0]["sweep_values"], out[1]["sweep_values"])
    assert out[0]["opt_logl"] == out[1]["opt_logl"]


def test_hbl_spmd_two_host_processes_with_the_host_side_exchange(tmp_path):
    """r06: SPMD site sharding with HYPHY_HIP_COLLECTIVE=host — two REAL host processes run the same batch file, each holds half of
    the patterns in a partition of its own, every evaluation ends in ONE shared-memory exchange of the two partial log-likelihoods
    (hyphy_hip_evaluate(_built)_exchange; no RCCL, so both processes may share device 0 and a one-GPU box runs it): LFCompute, an
    R sweep and a full Optimize report the whole alignment's values, identical on both ranks."""
    _need_binaries()
    from oracle import hbl
    import threading
    fx = common.load("codon_wide")
    case = _case("codon", 64, 60, 15)
    out = {}

    def run(rank):
        env = dict(ENV, HYPHY_HIP_WORLD="2", HYPHY_HIP_RANK=str(rank), HYPHY_HIP_LOCAL_RANK="0", HYPHY_HIP_COLLECTIVE="host",
                   HYPHY_HIP_RUN_ID=f"it{os.getpid()}_{id(tmp_path) & 0xffff}")
        out[rank] = hbl.evaluate(binary=HIP_BIN, extra_env=env, per_site=False, optimize=True,
                                 sweep=dict(param="R", start=0.3, step=0.01, n=8, record=8), **case)

    th = [threading.Thread(target=run, args=(r,)) for r in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=900)
    ref = float(fx["logl"])
    for r in (0, 1):
        assert "host-side exchange per evaluation" in out[r]["stdout"], out[r]["stdout"][-600:]
        assert abs(out[r]["logl"] - ref) <= 1e-10 * abs(ref), (r, out[r]["logl"], ref)
    assert np.array_equal(out[0]["sweep_values"], out[1]["sweep_values"])
    assert out[0]["opt_logl"] == out[1]["opt_logl"]
    cpu = hbl.evaluate(per_site=False, optimize=True, **case)
    assert abs(out[0]["opt_logl"] - cpu["opt_logl"]) <= 2e-3


def _run_constrained_local_model(binary=None, env=None, optimize=False, sweep=None, branch_specific=False):
    """MG94 written the way HyPhy's own codon models carry a global omega: two LOCAL parameters per branch (synRate,
    nonSynRate) and the constraint `givenTree.N.nonSynRate := R*givenTree.N.synRate` on every branch — nonSynRate is a
    DEPENDENT local.  branch_specific: every third branch is constrained through a second global R2 instead (foreground /
    background omega classes): the constraints then differ between branches."""
    import tempfile
    from hyphy_amd import data, models, tree
    from oracle import hbl, make_golden as mg
    syn = data.evolve(16, 60, 3, seed=31)
    bt = mg.branch_lengths(syn.flat, 38, 0.02, 0.12)
    lines = ["MGQ = {61,61};"]
    for (i, j, name, ns, pf) in models.mg94rev_template(mg.POS_FREQS):
        parts = ([name] if name != "AG" else []) + ["nonSynRate" if ns else "synRate", repr(float(pf))]
        lines.append(f"MGQ[{i}][{j}] := {'*'.join(parts)};")
    lines.append("vectorOfFrequencies = {\n" + ",\n".join("{" + repr(float(v)) + "}" for v in models.f3x4_codon_freqs(mg.POS_FREQS)) + "};")
    lines.append("Model MGM = (MGQ, vectorOfFrequencies, 0);")
    tmp = tempfile.mkdtemp(prefix="hydep_")
    fasta, outp = os.path.join(tmp, "aln.fasta"), os.path.join(tmp, "out.txt")
    hbl.write_fasta(fasta, syn.flat.leaf_names, syn.seqs)
    g = dict(R=0.3, **mg.REV)
    if branch_specific:
        g["R2"] = 0.3   # (equal at the start: the classes only separate once a parameter moves)
    txt = hbl.build_script(fasta=fasta, newick=tree.to_newick(syn.tree), unit=3, model_block="\n".join(lines), model_name="MGM",
                           globals_=g, branch_t=bt, out_path=outp, per_site=False, optimize=optimize, sweep=sweep)
    for k, (nm, t) in enumerate(bt.items()):
        om = "R2" if (branch_specific and k % 3 == 0) else "R"
        txt = txt.replace(f"givenTree.{nm}.t = {hbl._fmt(t)};",
                          f"givenTree.{nm}.synRate = {hbl._fmt(t)}; givenTree.{nm}.nonSynRate := {om}*givenTree.{nm}.synRate;")
    assert ".t = " not in txt
    out = hbl.run_script(txt, tmp, binary=binary, extra_env=env)
    res = hbl.parse_output(outp)
    res["stdout"] = out
    return res


def test_hbl_template_mode_with_constrained_dependent_locals():
    """r03: template mode also takes models whose branches carry DEPENDENT locals, provided every branch carries the same
    constraints.  Global-omega MG94 with `nonSynRate := R*synRate` per branch: a sweep of R in mode B and a
    complete Optimize through template mode (K = 1) against the unmodified binary.  (Branch-specific constraints: the next test.)"""
    _need_binaries()
    sweep = dict(param="R", start=0.3, step=0.02, n=15, record=15)
    envB = dict(ENV, HYPHY_HIP_DEVICE_EXPM="always")
    cpu = _run_constrained_local_model(sweep=sweep)
    gpu = _run_constrained_local_model(binary=HIP_BIN, env=envB, sweep=sweep)
    n_eval, K, n_skip = _template_evals(gpu["stdout"])
    assert K == 1 and n_eval >= 10 and n_skip > 0, gpu["stdout"][-1000:]
    assert np.max(np.abs(gpu["sweep_values"] - cpu["sweep_values"]) / np.abs(cpu["sweep_values"])) < 1e-10
    cpu = _run_constrained_local_model(optimize=True)
    gpu = _run_constrained_local_model(binary=HIP_BIN, env=ENV, optimize=True)
    assert _template_evals(gpu["stdout"])[0] > 5, gpu["stdout"][-1000:]
    assert abs(gpu["opt_logl"] - cpu["opt_logl"]) <= 2e-3


def test_hbl_template_mode_with_branch_classes():
    """r04: foreground / background omega — how FEL, BUSTED and RELAX partition their branches (calcnode.cpp:526-704 evaluates the
    dependents branch by branch).  Every third branch is constrained through R2, the others through R: two constraint signatures,
    i.e. two branch classes with templates of their own (one coefficient column each).  Sweeps of either global (the classes are
    equal at the start and separate during the sweep: branches with EQUAL local parameters then carry DIFFERENT matrices) and a
    complete Optimize, in template mode, against the unmodified binary."""
    _need_binaries()
    envB = dict(ENV, HYPHY_HIP_DEVICE_EXPM="always")
    for param in ("R", "R2"):
        sweep = dict(param=param, start=0.3, step=0.02, n=15, record=15)
        cpu = _run_constrained_local_model(sweep=sweep, branch_specific=True)
        gpu = _run_constrained_local_model(binary=HIP_BIN, env=envB, sweep=sweep, branch_specific=True)
        n_eval, K, n_skip = _template_evals(gpu["stdout"])
        assert K == 2 and n_eval >= 10 and n_skip > 0, gpu["stdout"][-1500:]
        assert "2 branch class(es)" in gpu["stdout"]
        assert np.max(np.abs(gpu["sweep_values"] - cpu["sweep_values"]) / np.abs(cpu["sweep_values"])) < 1e-10
    cpu = _run_constrained_local_model(optimize=True, branch_specific=True)
    gpu = _run_constrained_local_model(binary=HIP_BIN, env=ENV, optimize=True, branch_specific=True)
    assert _template_evals(gpu["stdout"])[0] > 5, gpu["stdout"][-1500:]
    assert abs(gpu["opt_logl"] - cpu["opt_logl"]) <= 2e-3
    # dense mode B stays available and agrees (HYPHY_HIP_TEMPLATES=0)
    sweep = dict(param="R2", start=0.3, step=0.02, n=6, record=6)
    cpu = _run_constrained_local_model(sweep=sweep, branch_specific=True)
    gpu = _run_constrained_local_model(binary=HIP_BIN, env=dict(envB, HYPHY_HIP_TEMPLATES="0"), sweep=sweep, branch_specific=True)
    assert _template_evals(gpu["stdout"])[0] == 0 and _device_calls(gpu["stdout"]) > 4
    assert np.max(np.abs(gpu["sweep_values"] - cpu["sweep_values"]) / np.abs(cpu["sweep_values"])) < 1e-10
