#!/usr/bin/env python3
"""Merge rocprofv3 --pmc passes (one counter group per pass, as MI355X_MICROARCH.md prescribes) into
profiles/<tag>_pmc_joint_8192.csv and profiles/<tag>_traffic.json.

    python3 scripts/pmc_summary.py <tag> <dir_with_pass_subdirs>

Each pass directory holds a *counter_collection.csv of `python3 scripts/prof_run.py 8192 3`.  Per kernel the LAST
dispatch (steady state) is kept.  HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB counters; gfx950 counts
64 B per 128-B request on wide coalesced reads, hence the factor 2), summed per kernel group of rfs_kernel_id."""
import csv, glob, json, os, sys

GROUPS = [("k_prep", "prep"), ("k_rf_passA", "rf_pass_a"), ("k_rf_mid", "rf_mid"), ("k_rf_passB", "rf_pass_b"),
          ("k_swd_roots", "swd_roots"), ("k_swd_eigen", "swd_eigen"), ("k_rf_reduce", "combine"),
          ("k_swd_combine", "combine")]


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("rfs::", "")
    return n


def main():
    tag, root = sys.argv[1], sys.argv[2]
    data = {}      # kernel -> counter -> (value of last dispatch, dispatches)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        last = {}
        for r in rows:
            k = short(r["Kernel_Name"]); cn = r["Counter_Name"]
            key = (k, cn)
            did = int(r["Dispatch_Id"])
            if key not in last or did >= last[key][0]:
                # several rows per dispatch (one per XCD / dimension): sum them
                if key in last and did == last[key][0]:
                    last[key] = (did, last[key][1] + float(r["Counter_Value"]), last[key][2])
                else:
                    last[key] = (did, float(r["Counter_Value"]), last.get(key, (0, 0, 0))[2] + 1)
        for (k, cn), (did, v, nd) in last.items():
            data.setdefault(k, {})[cn] = (v, nd)
    counters = sorted({c for k in data for c in data[k]})
    os.makedirs("profiles", exist_ok=True)
    out = os.path.join("profiles", f"{tag}_pmc_joint_8192.csv")
    with open(out, "w") as fo:
        fo.write("# rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 scripts/prof_run.py 8192 3   (one pass per counter group)\n")
        fo.write("# per-dispatch values of the LAST (steady-state) dispatch of each kernel; FETCH_SIZE / WRITE_SIZE in KiB as reported\n")
        fo.write("kernel,dispatches," + ",".join(counters) + "\n")
        for k in sorted(data):
            if not k.startswith("k_"):
                continue
            nd = max(v[1] for v in data[k].values())
            fo.write(k + f",{nd}," + ",".join("%g" % data[k][c][0] if c in data[k] else "" for c in counters) + "\n")
    traffic = {"_comment": "HBM bytes per launch of each kernel group at config 2 (8192 chains), from " + out +
               ": 2*FETCH_SIZE (gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE; rocFFT kernels not included",
               "chains": 8192}
    for k in data:
        for pat, grp in GROUPS:
            if k.startswith(pat) and "FETCH_SIZE" in data[k] and "WRITE_SIZE" in data[k]:
                traffic[grp] = traffic.get(grp, 0.0) + (2 * data[k]["FETCH_SIZE"][0] + data[k]["WRITE_SIZE"][0]) * 1024.0
    json.dump(traffic, open(os.path.join("profiles", f"{tag}_traffic.json"), "w"), indent=1)
    print(open(out).read()); print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
