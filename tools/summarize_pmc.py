#!/usr/bin/env python3
"""Turns rocprofv3 --pmc counter_collection CSVs into the summaries kept under profiles/.

  traffic:  python tools/summarize_pmc.py traffic FETCH.csv WRITE.csv --out profiles/r02_traffic.json --precision split_f16 --batch 256
            per-launch HBM-side bytes of the step kernels (MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE in
            KiB, separate passes; on gfx950 FETCH_SIZE reports half of a wide coalesced streaming read -> doubled;
            WRITE_SIZE exact), stamped with the digest of the kernel sources it was captured on
  mfma:     python tools/summarize_pmc.py mfma COUNTERS.csv --out profiles/r02_x_pmc_mfma.csv
            SQ_VALU_MFMA_BUSY_CYCLES per SIMD-cycle (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), per kernel
"""
import argparse
import csv
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# kernel symbol -> the step-kernel names bench.py uses
STEP_KERNELS = [
    ("frame_lstm_kernel", "prenet+lstm_att"), ("attn_lstm_kernel", "attention+lstm_dec"), ("lstm_lean_kernel", "lstm_lean(alone)"),
    ("frame_kernel", "prenet"), ("attn_kernel", "attention"), ("proj_kernel", "proj"),
]


STEP_LABELS = {label for _, label in STEP_KERNELS} | {"lstm_att", "lstm_dec", "query", "proj+prenet+lstm_att", "query+attention+lstm_dec"}
# the step order the capture ran (tools/prof_kernels.py prints it): the two-role kernels carry their roles' names from it
ORDER_NAMES = []


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if "frame_lstm_kernel" in name and name.rstrip().endswith(", true>(ttsdec::FrameArgs, ttsdec::LstmArgs, ttsdec::ProjArgs, int, int, int, int)"):
        return "proj+prenet+lstm_att"  # (the three-role form of the frame launch)
    if "attn_lstm_kernel" in name and "query+attention+lstm_dec" in ORDER_NAMES:
        return "query+attention+lstm_dec"  # (the attention role's workgroups also run the query GEMM: same symbol)
    for key, label in STEP_KERNELS:
        if key in name:
            return label
    if "lstm_kernel" in name:
        return "lstm_dec" if name.rstrip().endswith("1>(ttsdec::LstmArgs)") else "lstm_att"
    if "gemm_rows_kernel<ttsdec::TileCfg<1, 1, 2, 4, 1, 0, 1, 1, 0>, 0, 0>" in name or "gemm_rows_kernel<ttsdec::TileCfg<1, 1, 4, 4, 0, 0, 1, 1, 0>, 0, 0>" in name:
        return "query"  # (the split-K query projection is the only user of these instantiations - split-fp16 / exact fp32 - in a decode step)
    if "gemm_rows_kernel" in name:
        return "gemm_rows:" + name.split("gemm_rows_kernel<")[1].split(">(")[0].replace("ttsdec::", "")
    return name.split("(")[0][:80]


def read(path):
    per = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> values per dispatch
    dur = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            per[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur[(k, row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    return per, dur


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["traffic", "mfma"])
    ap.add_argument("files", nargs="+")
    ap.add_argument("--out", required=True)
    ap.add_argument("--precision", default="split_f16")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--note", default="")
    ap.add_argument("--order", default="", help="comma-separated launch names of the captured step order (prof_kernels.py's output keys)")
    a = ap.parse_args()
    ORDER_NAMES.extend(n for n in a.order.split(",") if n)
    if a.mode == "traffic":
        import bench

        fetch, _ = read(a.files[0])
        write, _ = read(a.files[1])
        per_launch = {}
        for k in fetch:
            fs = fetch[k].get("FETCH_SIZE", [])
            ws = write.get(k, {}).get("WRITE_SIZE", [])
            if not fs or not ws or k not in STEP_LABELS:
                continue
            f_kib, w_kib = sum(fs) / len(fs), sum(ws) / len(ws)
            per_launch[k] = {"launches": len(fs), "FETCH_SIZE_KiB": round(f_kib, 1), "WRITE_SIZE_KiB": round(w_kib, 1),
                             "hbm_bytes": int(2 * f_kib * 1024 + w_kib * 1024)}
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or None
        json.dump({
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/prof_kernels.py "
                      f"--precision {a.precision} --batch {a.batch} (L=120); " + a.note,
            "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced "
                          "(16 B/lane) streaming read, LDS-DMA included -> doubled; WRITE_SIZE is exact; both counters are in KiB",
            "sources_digest": bench.sources_digest(), "captured_at_commit": commit, "precision": a.precision, "batch": a.batch,
            "per_launch": per_launch}, open(a.out, "w"), indent=1)
        print(json.dumps(per_launch, indent=1))
    else:
        per, _ = read(a.files[0])
        rows = []
        for k, c in per.items():
            busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", []), c.get("GRBM_GUI_ACTIVE", [])
            if not busy or not act:
                continue
            n = len(busy)
            b, g = sum(busy) / n, sum(act) / n
            rows.append((k, n, g, b, b / (g / 8 * 1024) if g else 0.0))
        rows.sort(key=lambda r: -r[2] * r[1])
        with open(a.out, "w") as f:
            f.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; per-launch averages. " + a.note + "\n")
            f.write("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)\n")
            f.write("kernel,launches,gpu_cycles,SQ_VALU_MFMA_BUSY_CYCLES,mfma_util\n")
            for r in rows[:24]:
                f.write(f"\"{r[0]}\",{r[1]},{r[2]:.0f},{r[3]:.0f},{r[4]:.4f}\n")
        print(open(a.out).read())


if __name__ == "__main__":
    main()
