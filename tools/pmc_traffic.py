#!/usr/bin/env python3
"""profiles/r0N_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes as
MI355X_MICROARCH.md prescribes): per-launch averages per kernel, FETCH_SIZE doubled (gfx950 counts 64 B per
128-B request), bytes = fetch*2 + write.   python tools/pmc_traffic.py <fetch_csv> <write_csv> <out_json> <note>"""
import collections, csv, json, re, sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            m = re.search(r"(k_\w+(<[^>]*>)?)", row["Kernel_Name"])
            if m:
                acc[m.group(1)].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    fetch_csv, write_csv, out, note = sys.argv[1:5]
    fetch, nf = per_kernel(fetch_csv, "FETCH_SIZE")
    write, _ = per_kernel(write_csv, "WRITE_SIZE")
    # rocprofv3 reports both in kB on this stack? calibrate: k_net44_forward reads 8.0 MB of centroids per launch
    kernels = {}
    for k in sorted(fetch):
        fr, wr = fetch[k], write.get(k, 0.0)
        kernels[k] = {"launches": nf[k], "fetch_raw": fr, "write_raw": wr}
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB on this stack (calibrated in round 1 on the forward kernel's
    # 8.0 MB of centroids: profiles/r01_traffic.json, counter_unit_bytes 1024)
    unit = 1024.0
    total = 0.0
    for k, v in kernels.items():
        v["fetch_bytes_raw"] = v.pop("fetch_raw") * unit
        v["write_bytes"] = v.pop("write_raw") * unit
        v["hbm_bytes_corrected"] = 2.0 * v["fetch_bytes_raw"] + v["write_bytes"]
    per_iter = {k: v["hbm_bytes_corrected"] for k, v in kernels.items()}
    json.dump({"note": note, "counter_unit_bytes": unit, "kernels": kernels}, open(out, "w"), indent=1)
    print(json.dumps(per_iter, indent=1))


if __name__ == "__main__":
    main()
