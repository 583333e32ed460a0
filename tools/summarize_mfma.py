"""MFMA utilisation per kernel from a rocprofv3 `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass.

SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over every SIMD (= 32 x the number of
v_mfma_f32_32x32x16 issued); GRBM_GUI_ACTIVE is reported as the sum over the 8 XCDs, so kernel cycles =
GRBM_GUI_ACTIVE / 8 (MI355X_MICROARCH.md, DVFS note).  utilisation = busy / (kernel cycles x 1024 SIMDs).
usage: summarize_mfma.py counter_collection.csv out.json [kernel-name substring ...]
"""
import collections
import csv
import json
import sys

SIMDS = 256 * 4


def main(path, out, *subs):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if subs and not any(s in name for s in subs):
            continue
        per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for name, c in per.items():
        busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES"), c.get("GRBM_GUI_ACTIVE")
        if not busy or not act:
            continue
        cyc = sum(act) / len(act) / 8.0
        b = sum(busy) / len(busy)
        res[name.split("(")[0][-90:]] = {"dispatches": len(busy), "mfma_busy_cycles": b, "kernel_cycles": cyc,
                                         "mfma_utilisation": round(b / (cyc * SIMDS), 4)}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        print(f"{v['mfma_utilisation']:.3f}  {v['dispatches']:4d}x  {k}")


if __name__ == "__main__":
    main(*sys.argv[1:])
