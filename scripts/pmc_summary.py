#!/usr/bin/env python3
"""Merge rocprofv3 --pmc passes (one counter group per pass, as MI355X_MICROARCH.md prescribes) into
profiles/<tag>_pmc_config<c>.csv and the per-step, per-group figures bench.py reads: profiles/<tag>_counters.json.

    python3 scripts/pmc_summary.py <tag> <config> <steps> <dir_with_pass_subdirs>

Each pass directory holds a *counter_collection.csv of `python3 scripts/prof_run.py <config> <steps> hmc` (a sampler run
that continues burned-in chains).  Per kernel: the sum over all dispatches divided by the device steps of the pass (the
dispatches of k_flow_post); "swd_roots_full_search" = the searches of the start models and of handed-back chains.  HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE
(KiB counters; gfx950 counts 64 B per 128-B request on wide coalesced reads, hence the factor 2)."""
import csv, glob, json, os, sys

GROUPS = [("k_prep", "prep"), ("k_rf_passA", "rf_pass_a"), ("k_rf_mid", "rf_mid"), ("k_rf_passB", "rf_pass_b"),
          ("k_swd_roots", "swd_roots_full_search"), ("k_swd_warm", "swd_roots"), ("k_swd_exact", "swd_exact"),
          ("k_swd_eigen", "swd_eigen"),
          ("k_rf_reduce", "rf_pass_b"), ("k_swd_combine", "combine"), ("k_flow_post", "combine")]


def short(name):
    return name.split("(")[0].replace("void ", "").replace("rfs::", "")


def main():
    tag, config, steps, root = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    tot, disp = {}, {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        seen = {}
        for r in csv.DictReader(open(f)):
            k, cn = short(r["Kernel_Name"]), r["Counter_Name"]
            tot[(k, cn)] = tot.get((k, cn), 0.0) + float(r["Counter_Value"])
            seen.setdefault((k, cn), set()).add(r["Dispatch_Id"])
        for key, ids in seen.items():
            disp[key] = len(ids)
    kernels = sorted({k for k, _ in tot if k.startswith("k_")})
    counters = sorted({c for _, c in tot})

    # device steps of the pass = dispatches of k_flow_post (one per flow step; k_prep_joint also runs in the plain evaluations of
    # HMCDualAveraging._find_initial_dt, which are not steps -- ADVICE r04); passes without a flow step fall back to k_prep_joint
    nstep = max([disp.get(("k_flow_post", c), 0) for c in counters] + [0])
    if nstep == 0:
        nstep = max([disp.get(("k_prep_joint", c), 0) for c in counters] + [1])
    steps = nstep

    def calls(k):
        return nstep
    os.makedirs("profiles", exist_ok=True)
    out = os.path.join("profiles", f"{tag}_pmc_config{config}.csv")
    with open(out, "w") as fo:
        fo.write(f"# rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 scripts/prof_run.py {config} {steps} hmc   (one pass per counter group)\n")
        fo.write("# per LEAPFROG STEP: sum over all dispatches of the kernel / calls it ran in; FETCH_SIZE / WRITE_SIZE in KiB as reported\n")
        fo.write("kernel,dispatches_per_step," + ",".join(counters) + "\n")
        for k in kernels:
            nd = max(disp.get((k, c), 0) for c in counters)
            fo.write(k + f",{nd / calls(k):g}," + ",".join("%g" % (tot[(k, c)] / calls(k)) if (k, c) in tot else "" for c in counters) + "\n")
    jpath = os.path.join("profiles", f"{tag}_counters.json")
    J = json.load(open(jpath)) if os.path.exists(jpath) else {
        "_comment": "per leapfrog step and kernel group, 8192 chains, from profiles/" + tag + "_pmc_config*.csv: hbm_bytes = "
                    "2*FETCH_SIZE (gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE; "
                    "valu_insts = SQ_INSTS_VALU (wave-instructions); rocFFT kernels not included", "chains": 8192}
    sec = {}
    for k in kernels:
        for pat, grp in GROUPS:
            if k.startswith(pat):
                g = sec.setdefault(grp, {"hbm_bytes": 0.0, "valu_insts": 0.0})
                if (k, "FETCH_SIZE") in tot and (k, "WRITE_SIZE") in tot:
                    g["hbm_bytes"] += (2 * tot[(k, "FETCH_SIZE")] + tot[(k, "WRITE_SIZE")]) * 1024.0 / calls(k)
                if (k, "SQ_INSTS_VALU") in tot:
                    g["valu_insts"] += tot[(k, "SQ_INSTS_VALU")] / calls(k)
                # FP64 vector flops (round 6): wave-instructions by class, FMA = 2 flops, ADD / MUL / TRANS = 1, x 64 lanes
                f64 = [tot.get((k, "SQ_INSTS_VALU_" + cls + "_F64")) for cls in ("ADD", "MUL", "FMA", "TRANS")]
                if all(v is not None for v in f64):
                    g["fp64_flops"] = g.get("fp64_flops", 0.0) + (f64[0] + f64[1] + 2 * f64[2] + f64[3]) * 64.0 / calls(k)
                    g["fp64_wave_insts"] = g.get("fp64_wave_insts", 0.0) + sum(f64) / calls(k)
                break
    J[f"config{config}"] = sec
    json.dump(J, open(jpath, "w"), indent=1)
    print(open(out).read()); print(json.dumps(sec, indent=1))


if __name__ == "__main__":
    main()
