"""Which uninitialised device buffer does a stress case depend on?  (diagnostic; GPU box)
python tests/poison_bisect.py <seed> : runs tests/stress_codon.py 1 <seed> with HYPHY_HIP_POISON=1, then once per buffer
with that buffer's poisoning switched off."""
import os, subprocess, sys
seed = sys.argv[1]
bufs = "codes codes_tile bc_ops bc_prog bc_slot bc_q freq pin ambig partials counts site_lik site_cnt mixed_lik mixed_cnt Prow Pfrag PTg qbuf slots ops prog frag_ctr hand_cnt pi out status weights wg_sum wg_cnt wg_flag".split()
def run(extra):
    env = dict(os.environ, HYPHY_HIP_POISON="1", **extra)
    r = subprocess.run([sys.executable, "tests/stress_codon.py", "1", seed], env=env, capture_output=True, text=True, timeout=300)
    return (r.stdout + r.stderr).strip().splitlines()[-1]
print("all poisoned:", run({}))
for b in bufs:
    out = run({"HYPHY_HIP_NOPOISON_s." + b: "1"})
    if "MISMATCH" not in out:
        print("clean without poisoning", b, ":", out)
