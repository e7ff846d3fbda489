#!/usr/bin/env python3
"""Instruction mix of the innermost loops of the library's kernels (gfx950 ISA from `hipcc -S --cuda-device-only`):
VALU count, v_cndmask reading VCC (e32: ~20 clocks each on gfx950, scripts/select_cost.hip) against SGPR-pair masks (e64: ~5.5),
SGPR spill reloads (v_readlane), f64 compares.

    python3 scripts/loop_mix.py [lib.s] [kernel-name-substring ...]
"""
import re, subprocess, sys, os

def isa(path):
    if not os.path.exists(path):
        here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rfsurfhmc_amd", "csrc")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-Wno-unused-value", os.path.join(here, "rfsurf_hip.hip"), "-o", path], check=True,
                       stderr=subprocess.DEVNULL)
    return open(path).read()

def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/lib.s"
    pats = sys.argv[2:] or ["k_rf_passA", "k_rf_mid_fused", "k_rf_passB", "k_swd_warmI", "k_swd_warm_check", "k_swd_exact",
                            "k_swd_eigen", "k_swd_warm_walk"]
    txt = isa(path)
    for f in re.split(r'\n(?=_Z\w+:\s)', txt):
        m = re.match(r'(_Z\w+):', f)
        if not m or not any(p in m.group(1) for p in pats):
            continue
        lines = f.split('s_endpgm')[0].split('\n')
        heads = [i for i, l in enumerate(lines) if 'Inner Loop Header' in l]
        out = []
        for h in heads:
            while h > 0 and not lines[h].startswith('.LBB'):      # (the comment may sit on the line after the label)
                h -= 1
            lab = lines[h].split(':')[0].strip()
            # the loop = its header block and every block annotated "in Loop: Header=<lab>" (rotated loops branch back to one of
            # those, not to the header), up to the end of the last of them
            tag = 'Header=' + lab[2:] + ' '
            blocks = [h] + [i for i, l in enumerate(lines) if l.startswith('.LBB') and tag in l + ' ']
            lo, hi = min(blocks), max(blocks)
            hi = next((i for i in range(hi + 1, len(lines)) if lines[i].startswith('.LBB')), len(lines)) - 1
            body = lines[lo:hi + 1]
            valu = sum(1 for l in body if re.match(r'\s+v_', l))
            if valu < 60:
                continue
            c = lambda rx: sum(1 for l in body if re.match(r'\s+' + rx, l))
            out.append((valu, c('v_cndmask_b32_e32'), c('v_cndmask_b32_e64'), c('v_readlane'), c(r'v_cmp\w*_f64'),
                        c('v_fma_f64|v_fmac_f64|v_mul_f64|v_add_f64'), c('v_mov_b'), c('scratch_')))
        if out:
            print(m.group(1)[:90])
            for o in out:
                print("   loop: VALU %4d | cndmask vcc %3d  sgpr %3d | readlane %3d | cmp_f64 %3d | f64 arith %4d | v_mov %3d | scratch %d" % o)

if __name__ == "__main__":
    main()
