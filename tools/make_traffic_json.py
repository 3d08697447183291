"""profiles/<tag>/traffic.json from the FETCH_SIZE / WRITE_SIZE tables of tools/collect_profiles.sh:
HBM bytes per launch of the GEMM family = (2 x FETCH_SIZE + WRITE_SIZE) KB summed over its kernels / dispatches
(MI355X_MICROARCH.md: separate --pmc passes; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950).
usage: python tools/make_traffic_json.py profiles/r02
"""
import json
import os
import sys


def table(path, counter):
    rows = {}
    for line in open(path):
        c = [x.strip() for x in line.strip().strip("|").split("|")]
        if len(c) == 5 and c[1] == counter:
            rows[c[0]] = (int(c[2]), float(c[3]))
    return rows


def main():
    d = sys.argv[1]
    fetch = table(os.path.join(d, "pmc_fetch.md"), "FETCH_SIZE")
    write = table(os.path.join(d, "pmc_write.md"), "WRITE_SIZE")
    names = [k for k in fetch if k.startswith("gemm_") and "x3" not in k]
    n = sum(fetch[k][0] for k in names)
    f = sum(fetch[k][1] for k in names) / n
    w = sum(write[k][1] for k in names if k in write) / sum(write[k][0] for k in names if k in write)
    out = {
        "kernel": "f32-MFMA GEMM (" + ", ".join("%s x%d" % (k, fetch[k][0]) for k in names) + ")",
        "source": "%s/pmc_fetch.md + pmc_write.md (rocprofv3 --pmc, separate passes, command in %s/command.txt)" % (d, d),
        "dispatches": n,
        "FETCH_SIZE_KB_per_launch": f,
        "WRITE_SIZE_KB_per_launch": w,
        "gfx950_fetch_correction": 2.0,
        "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
        "note": "2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by 2x on "
                "gfx950); stream-K launches hand partial tiles over through sc1 accesses, counted here",
    }
    with open(os.path.join(d, "traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    # the package-default form's dominant kernel (its own PMC passes: pmc_fetch_x3.md / pmc_write_x3.md)
    fx, wx = os.path.join(d, "pmc_fetch_x3.md"), os.path.join(d, "pmc_write_x3.md")
    if os.path.exists(fx) and os.path.exists(wx):
        fetch, write = table(fx, "FETCH_SIZE"), table(wx, "WRITE_SIZE")
        names = [k for k in fetch if k.startswith("gemm_x3")]
        if names:
            n = sum(fetch[k][0] for k in names)
            f = sum(fetch[k][1] for k in names) / n
            w = sum(write[k][1] for k in names if k in write) / max(1, sum(write[k][0] for k in names if k in write))
            with open(os.path.join(d, "traffic_x3.json"), "w") as fh:
                json.dump({"kernel": "3 x bf16 split GEMM (" + ", ".join("%s x%d" % (k, fetch[k][0]) for k in names) + ")",
                           "source": "%s/pmc_fetch_x3.md + pmc_write_x3.md (command in %s/command_x3.txt)" % (d, d),
                           "dispatches": n, "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
                           "gfx950_fetch_correction": 2.0, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
