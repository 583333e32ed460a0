"""Turn gpurun_out/<tag>/ (tools/profile_round.sh) into the committed summaries profiles/<tag>_*.
usage: python tools/collect_profiles.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(tag, sub, pattern):
    hits = glob.glob(os.path.join(ROOT, "gpurun_out", tag, sub, "**", pattern), recursive=True)
    return hits[0] if hits else None


def ours(name):
    return "mmrag" in name


def main(tag):
    out = os.path.join(ROOT, "profiles")
    src = os.path.join(ROOT, "gpurun_out", tag)
    for f, dst in (("bench.json", f"{tag}_bench.json"),):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(out, dst))
    for sub, dst in (("stats", f"{tag}_kernel_stats.csv"), ("embed_stats", f"{tag}_embed_kernel_stats.csv"),
                     ("embed_other_stats", f"{tag}_embed_other_shapes_kernel_stats.csv")):
        f = find(tag, sub, "*kernel_stats.csv")
        if f:
            rows = [r for r in csv.reader(open(f))]
            keep = [rows[0]] + [r for r in rows[1:] if ours(r[0])]      # this library's kernels only
            csv.writer(open(os.path.join(out, dst), "w", newline="")).writerows(keep)
            print(dst, len(keep) - 1, "kernels")

    # HBM traffic per scan: all dispatches of the search kernels / number of scans (one qs_seed_thr_kernel per scan,
    # else one merge_topk_kernel pair ...), gfx950 correction applied to FETCH_SIZE
    def per_scan(sub, counter):
        f = find(tag, sub, "*counter_collection.csv")
        if not f:
            return None, 0
        tot, walks, seeds = 0.0, 0, 0
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or not ours(r["Kernel_Name"]):
                continue
            if "cosine_topk" in r["Kernel_Name"] or "qs_seed" in r["Kernel_Name"]:
                tot += float(r["Counter_Value"])
            walks += "cosine_topk_walk_kernel" in r["Kernel_Name"]      # single-launch plan: one per scan
            seeds += "qs_seed_thr_kernel" in r["Kernel_Name"]           # three-launch plan: one per scan
        scans = walks or seeds
        return (tot / scans if scans else None), scans

    fetch, n1 = per_scan("pmc_fetch", "FETCH_SIZE")
    write, n2 = per_scan("pmc_write", "WRITE_SIZE")
    if fetch is not None:
        t = {"kernel": "the search kernels of one scan (cosine_topk_walk_kernel; before round 3: sample pass + "
                       "qs_seed_thr_kernel + walk)", "rows_per_gpu": 1000000,
             "batch": 256, "scans": n1, "fetch_size_kib_raw": fetch, "hbm_read_bytes_per_launch": fetch * 1024 * 2,
             "correction": "FETCH_SIZE KiB x1024 x2 (gfx950 counts 128-B requests at 64 B)",
             "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, round {tag}"}
        total = t["hbm_read_bytes_per_launch"]
        if write is not None:
            t.update(write_size_kib_raw=write, hbm_write_bytes_per_launch=write * 1024)
            total += write * 1024
        t["hbm_bytes_per_launch"] = total
        json.dump(t, open(os.path.join(out, "traffic.json"), "w"), indent=1)
        json.dump(t, open(os.path.join(out, f"{tag}_traffic.json"), "w"), indent=1)
        print("traffic per scan: %.3f GB (algorithmic 1.536)" % (total / 1e9))
    for sub, dst in (("pmc_mfma", f"{tag}_mfma_util_search.json"), ("embed_mfma", f"{tag}_mfma_util_embed.json")):
        f = find(tag, sub, "*counter_collection.csv")
        if not f:
            continue
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if ours(r["Kernel_Name"]):
                per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        res = {}
        for name, c in per.items():
            busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES"), c.get("GRBM_GUI_ACTIVE")
            if not busy or not act or sum(busy) == 0:
                continue
            cyc = sum(act) / len(act) / 8.0
            b = sum(busy) / len(busy)
            res[name.split("(")[0][-100:]] = {"dispatches": len(busy), "mfma_busy_cycles": b, "kernel_cycles": cyc,
                                              "mfma_utilisation": round(b / (cyc * 1024), 4)}
        json.dump(res, open(os.path.join(out, dst), "w"), indent=1)
        for k, v in sorted(res.items(), key=lambda kv: -kv[1]["mfma_busy_cycles"] * kv[1]["dispatches"])[:6]:
            print(f"  {v['mfma_utilisation']:.3f} {v['dispatches']:5d}x {k}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03")
