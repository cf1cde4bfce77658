"""Per-kernel register / LDS / occupancy table from hipcc's kernel-resource-usage remarks.
usage: python tools/resources.py [substring ...]   (run from anywhere; compiles both translation units to /dev/null)"""
import os, re, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "compressedsensing.jl_amd", "csrc")
flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c -o /dev/null".split()
rows, cur = [], None
for src in ("csmp.hip", "csmp_screen.hip"):
    out = subprocess.run(["/opt/rocm/bin/hipcc", *flags, src], cwd=root, capture_output=True, text=True).stderr
    for line in out.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+: +(\w[\w /\[\]]*): +(\S+)", line) or re.search(r": +(Function Name|Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): +(\S+)", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k in ("Function Name", "Name"):
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
pats = sys.argv[1:]
for r in rows:
    if pats and not any(p in r["name"] for p in pats):
        continue
    print(f'{r["name"][:90]:90s} vgpr {r.get("VGPRs","?"):>4} agpr {r.get("AGPRs","?"):>4} sgpr {r.get("TotalSGPRs","?"):>4} '
          f'scratch {r.get("ScratchSize [bytes/lane]","?"):>4} occ {r.get("Occupancy [waves/SIMD]","?"):>2} lds {r.get("LDS Size [bytes/block]","?")}')
