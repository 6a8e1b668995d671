#!/usr/bin/env python3
"""Summarise rocprofv3 `--pmc ... --output-format csv` passes into profiles/kernel_counters.json (read by bench.py).

    python tools/pmc_summary.py --workload c4 --out profiles/kernel_counters.json --csv-dir profiles/r02  pass1/p_counter_collection.csv ...

For every kernel of the library it averages each counter over the launches of the run (rocprofv3 sums a counter over the chip per
dispatch), writes the per-launch averages, and derives:
  valu_issue_slots  VALU issue time in units of 2 cycles (= one fp32 FMA slot): fp32 add / mul / fma and int32 instructions count 1,
                    every other VALU instruction (fp64, compare, select, min / max / med3, 64-bit integer) counts 2 -- the rates
                    tools/valu_rate.hip measures on gfx950; matrix instructions are not VALU issue slots and are removed
  valu_busy         2 cycles x valu_issue_slots / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)   (utilisation under the profiler's clocks)
  mfma_flops        512 x SQ_INSTS_VALU_MFMA_MOPS_F64
  hbm_bytes         FETCH_SIZE x 1024 (x 2 only for the kernels listed in WIDE_STREAMS, the guide's gfx950 correction applies to wide
                    coalesced streams) + WRITE_SIZE x 1024
It also writes one small CSV per input pass into --csv-dir (kernel, launches, counter averages) so a reader can redo the arithmetic.
"""
import argparse
import csv
import json
import os
import re
import sys
from collections import defaultdict

WIDE_STREAMS = ()  # kernels whose reads are 16 B / lane coalesced streams (none of the hot kernels: scalar-cache and 4-8 B / lane reads)


def short(name):
    name = name.strip().strip('"')
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(.*\)$", "", name)  # drop the parameter list
    name = name.replace("gorio::ug::", "").replace("gorio::", "").replace("(anonymous namespace)::", "")
    return name


def read_pass(path):
    per = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> [per-dispatch values]
    disp = defaultdict(set)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            per[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            disp[k].add(row["Dispatch_Id"])
    return per, {k: len(v) for k, v in disp.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv", nargs="+")
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--out", default=None)
    ap.add_argument("--csv-dir", default=None)
    ap.add_argument("--skip-first", type=int, default=0, help="drop the first N launches of every kernel (warm-up step)")
    ap.add_argument("--note", default="")
    args = ap.parse_args()
    merged = defaultdict(dict)
    launches = {}
    for path in args.csv:
        per, nd = read_pass(path)
        rows = []
        for k, ctrs in per.items():
            if k.startswith("__amd") or "at::native" in k or "rccl" in k.lower():
                continue
            for c, vals in ctrs.items():
                v = vals[args.skip_first:] if len(vals) > args.skip_first else vals
                merged[k][c] = sum(v) / len(v)
            launches[k] = max(launches.get(k, 0), nd[k])
            rows.append((k, nd[k], {c: merged[k][c] for c in ctrs}))
        if args.csv_dir:
            os.makedirs(args.csv_dir, exist_ok=True)
            tag = os.path.basename(os.path.dirname(os.path.abspath(path)))
            names = sorted({c for _, _, d in rows for c in d})
            with open(os.path.join(args.csv_dir, f"{tag}_per_kernel.csv"), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["kernel", "launches"] + names)
                for k, n, d in sorted(rows):
                    w.writerow([k, n] + [f"{d.get(c, float('nan')):.6g}" for c in names])
    out = {}
    for k, c in merged.items():
        e = dict(c)
        e["launches_in_profile"] = launches[k]
        if "SQ_INSTS_VALU" in c:
            mfma = c.get("SQ_INSTS_MFMA", 0.0)
            valu = c["SQ_INSTS_VALU"] - mfma
            if "SQ_INSTS_VALU_ADD_F32" in c:
                # measured issue rates on gfx950 (tools/valu_rate.hip): fp32 add / sub / mul / fma and 32-bit integer / logic instructions
                # hold a SIMD for 2 cycles per wave64 instruction; fp64 arithmetic, compares, selects, min / max / med3 and 64-bit
                # integer instructions for 4.  The SQ counters classify the first group; everything else is priced at the slow rate.
                fast = sum(c.get(x, 0.0) for x in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_INT32"))
                fast = min(fast, valu)
                e["valu_fast_instructions"], e["valu_slow_instructions"] = fast, valu - fast
                e["valu_issue_slots"] = fast + 2.0 * (valu - fast)  # in units of 2 cycles: one fp32-FMA issue slot
            else:
                f64 = sum(c.get(x, 0.0) for x in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
                e["valu_issue_slots"] = valu + f64
        if "valu_issue_slots" in e and c.get("GRBM_GUI_ACTIVE"):
            e["valu_busy"] = 2.0 * e["valu_issue_slots"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)  # under the profiler's own clocks
        if "SQ_INSTS_VALU_MFMA_MOPS_F64" in c:
            e["mfma_flops"] = 512.0 * c["SQ_INSTS_VALU_MFMA_MOPS_F64"]
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            fetch = c.get("FETCH_SIZE", 0.0) * 1024.0 * (2.0 if k in WIDE_STREAMS else 1.0)
            e["fetch_bytes"], e["write_bytes"] = fetch, c.get("WRITE_SIZE", 0.0) * 1024.0
            e["hbm_bytes"] = fetch + e["write_bytes"]
        out[f"{k}:{args.workload}"] = e
    import hashlib

    def source_sha16():
        h = hashlib.sha256()
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "go-rio_amd", "csrc")
        for name in sorted(os.listdir(d)):
            if name.endswith((".hip", ".h")):
                h.update(name.encode())
                h.update(open(os.path.join(d, name), "rb").read())
        return h.hexdigest()[:16]

    doc = {"_source_sha16": source_sha16(), "_note": ("per-launch averages of rocprofv3 --pmc passes (separate passes per counter group, never combined with the trace domains), MI355X. " + args.note).strip(),
           "kernels": out}
    if args.out:
        old = {}
        if os.path.exists(args.out):
            try:
                old = json.load(open(args.out)).get("kernels", {})
            except Exception:
                old = {}
        old_doc = {}
        try:
            old_doc = json.load(open(args.out))
        except Exception:
            pass
        if old_doc.get("_source_sha16") == doc["_source_sha16"]:  # entries of other workloads collected from the SAME sources stay
            for k, e in old.items():
                if not k.endswith(":" + args.workload):
                    doc["kernels"].setdefault(k, e)
                elif k in doc["kernels"]:
                    for kk, vv in e.items():  # fields another tool merged in (tools/search_work.py)
                        doc["kernels"][k].setdefault(kk, vv)
        json.dump(doc, open(args.out, "w"), indent=1, sort_keys=True)
    else:
        json.dump(doc, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
