"""Turn a rocprofv3 --pmc FETCH_SIZE (and optionally WRITE_SIZE) counter_collection.csv into
profiles/traffic.json for bench.py's roofline.traffic.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is reported in KiB and counts
exactly half of the bytes of a wide coalesced streaming read (128-B requests tallied at 64 B),
so hbm_read_bytes = FETCH_SIZE * 1024 * 2.  WRITE_SIZE (KiB) is exact for 16-B/lane stores.
"""
import csv
import json
import sys


def main(fetch_csv, out_json, rows_per_gpu, batch, write_csv=None, kernel="cosine_topk_kernel"):
    def avg(path, counter):
        # one scan launches each instantiation once (sample pre-pass + main pass): average per
        # instantiation, then add them up -> bytes per scan
        per = {}
        for r in csv.DictReader(open(path)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
        if not per:
            return None, 0
        return sum(sum(v) / len(v) for v in per.values()), max(len(v) for v in per.values())

    f, nf = avg(fetch_csv, "FETCH_SIZE")
    out = {"kernel": kernel, "rows_per_gpu": int(rows_per_gpu), "batch": int(batch), "scans": nf,
           "fetch_size_kib_raw": f, "hbm_read_bytes_per_launch": f * 1024 * 2,
           "correction": "FETCH_SIZE KiB x1024 x2 (gfx950 counts 128-B requests at 64 B)"}
    total = out["hbm_read_bytes_per_launch"]
    if write_csv:
        w, nw = avg(write_csv, "WRITE_SIZE")
        if w is not None:
            out["write_size_kib_raw"] = w
            out["hbm_write_bytes_per_launch"] = w * 1024
            total += w * 1024
    out["hbm_bytes_per_launch"] = total
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main(*sys.argv[1:])
