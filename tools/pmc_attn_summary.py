#!/usr/bin/env python
"""Per-kernel SQ counter table from one or more rocprofv3 PMC passes (csv output) of the same command.

    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA \
              SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d DIR -o t -- python tools/attn_one.py
    python tools/pmc_attn_summary.py DIR [DIR2 ...] [--match attn]

Columns: launches; avg duration (dispatch timestamps of the PMC pass itself - profiled passes clock lower, see
MI355X_MICROARCH.md DVFS note); every counter as the per-launch mean; derived: VALU instructions per MFMA instruction,
MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz) (busy cycles = 32 per
v_mfma_f32_32x32x16_bf16, one pipe per SIMD), and the MFMA-implied TFLOP/s = MFMA instr x 64 lanes... (32*32*16*2 flop
each) / duration."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void\s+", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^()]*>)?)", name)
    return (m.group(1) if m else name)[:64]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = "attn"
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        args = [a for a in args if a != match]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    for d in args:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if match not in k:
                    continue
                a = agg[k][r["Counter_Name"]]
                a[0] += 1; a[1] += float(r["Counter_Value"])
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    t = dur[k]
                    t[0] += 1; t[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for k in sorted(agg):
        n, tot = dur[k]
        us = tot / n / 1e3
        c = {name: v[1] / v[0] for name, v in agg[k].items()}
        print(f"{k}: {n} launches, avg {us:.1f} us (profiled pass)")
        for name in sorted(c):
            print(f"    {name:28s} {c[name]:16.0f}")
        mf, va = c.get("SQ_INSTS_MFMA"), c.get("SQ_INSTS_VALU")
        if mf and va:
            print(f"    VALU (non-MFMA) per MFMA     {(va - mf) / mf:16.2f}   (SQ_INSTS_VALU counts the MFMAs too)" if va > mf else
                  f"    VALU per MFMA                {va / mf:16.2f}")
        if mf:
            # the 16-row-wave attention kernels (attn4_*) issue v_mfma_f32_16x16x32_bf16: half the flop of a 32x32x16
            small = "attn4" in k
            fl = 16384 if small else 32768
            print(f"    MFMA-implied TFLOP/s         {mf * fl / (us * 1e-6) / 1e12:16.1f}   ({'16x16x32' if small else '32x32x16'}x2 flop per instruction)")
        b = c.get("SQ_VALU_MFMA_BUSY_CYCLES")
        if b:
            print(f"    MFMA pipe utilisation        {b / (1024 * us * 1e-6 * 2.4e9):16.3f}   (busy cycles / (1024 SIMDs x t x 2.4 GHz))")
        w, wc = c.get("SQ_WAVES"), c.get("SQ_WAVE_CYCLES")
        if w and wc:
            print(f"    cycles per wave              {4 * wc / w:16.0f}   (SQ_WAVE_CYCLES counts quad-cycles)")


if __name__ == "__main__":
    main()
