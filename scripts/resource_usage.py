"""Register / LDS / occupancy table of the library's kernels from hipcc -Rpass-analysis=kernel-resource-usage output:
    hipcc ... -Rpass-analysis=kernel-resource-usage ... 2> ru.txt ; python3 scripts/resource_usage.py ru.txt [filter ...]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
keys = sys.argv[2:] or ["passB", "passA", "exact", "warm", "eigen", "mid_fused", "combine", "flow_post", "prep_joint"]
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)
for b in blocks[1:]:
    name = b.split('\n')[0].strip()
    def g(k):
        m = re.search(k + r': (\d+)', b); return int(m.group(1)) if m else -1
    if not any(k in name for k in keys):
        continue
    try:
        d = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except Exception:
        d = name
    d = re.sub(r'\(.*', '', d).replace('void rfs::', '')[:70]
    sc, oc, lds = g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')
    print(f"{d:70s} VGPR {g('VGPRs'):4d} AGPR {g('AGPRs'):3d} scratch {sc:4d} occ {oc} SGPR {g('SGPRs'):3d} LDS {lds}")
