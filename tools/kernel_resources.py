#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py hyphy_amd/csrc/prune.hip [filter-substring] [extra hipcc flags...]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
err = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True).stderr
cur = None
rows = []
for ln in err.split("\n"):
    m = re.search(r"remark: (?:\s*)([A-Za-z \[\]/]+): (.*?) \[-Rpass", ln)
    if not m:
        if "error" in ln:
            print(ln)
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        name = subprocess.run(["/usr/bin/c++filt", v], stdout=subprocess.PIPE, text=True).stdout.strip()
        name = re.sub(r"hyhip::\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        cur = {"name": name}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print(f"{'kernel':60s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>7s} {'LDS':>7s} {'occ':>4s}")
for r in rows:
    if flt and flt not in r["name"]:
        continue
    print(f"{r['name'][:60]:60s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} "
          f"{r.get('ScratchSize [bytes/lane]', '?'):>7s} {r.get('LDS Size [bytes/block]', '?'):>7s} "
          f"{r.get('Occupancy [waves/SIMD]', '?'):>4s}")
