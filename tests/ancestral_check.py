import os, sys, tempfile
sys.path.insert(0, '/root/repo')
from oracle import hbl
from tests.test_hyphy_integration import _case, HIP_BIN, ENV
case = _case("codon", 8, 40, 11)
def run(binary, env, mode):
    tmp = tempfile.mkdtemp(prefix="anc_")
    fasta = os.path.join(tmp, "aln.fasta"); outp = os.path.join(tmp, "out.txt"); ancp = os.path.join(tmp, "anc.txt")
    hbl.write_fasta(fasta, case["names"], case["seqs"])
    txt = hbl.build_script(fasta=fasta, newick=case["newick"], unit=case["unit"], model_block=case["model_block"],
                           model_name=case["model_name"], globals_=case["globals_"], branch_t=case["branch_t"],
                           out_path=outp, per_site=False)
    extra = {"joint": "DataSet anc = ReconstructAncestors (lf);",
             "marginal": "DataSet anc = ReconstructAncestors (lf, MARGINAL);",
             "sample": "SetParameter (RANDOM_SEED, 7, 0); DataSet anc = SampleAncestors (lf);"}[mode]
    txt += extra + f'\nDataSetFilter af = CreateFilter (anc, 1);\nDATA_FILE_PRINT_FORMAT = 9;\nfprintf ("{ancp}", CLEAR_FILE, af);\n'
    out = hbl.run_script(txt, tmp, binary=binary, extra_env=env)
    return open(ancp).read(), out
for mode in ("joint", "marginal"):
    cpu, _ = run(None, None, mode)
    gpu, so = run(HIP_BIN, ENV, mode)
    print(mode, "identical" if cpu == gpu else "DIFFERENT", len(cpu), len(gpu))
    if cpu != gpu:
        print(cpu[:300]); print(gpu[:300])
    print(so[-300:])
