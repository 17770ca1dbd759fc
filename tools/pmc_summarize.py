#!/usr/bin/env python
"""Turns the two rocprofv3 --pmc passes of tools/pmc_hbm.sh (FETCH_SIZE, WRITE_SIZE over tools/eager_forward.py) into
profiles/rNN_pmc_hbm_per_kernel.csv and profiles/rNN_hbm_traffic.json.

bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024: on gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced streaming read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-byte-per-lane stores.
The json records the sha256 of the kernel sources it was measured on: bench.py only quotes it while the sources match.

usage: python tools/pmc_summarize.py <pmc_outdir> <round tag, e.g. r02>
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "detectron2-centernet_amd", "csrc", "*"))):
        if f.endswith((".hip", ".h")):
            with open(f, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def per_kernel(path, counter):
    files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {path}"
    tot, n = defaultdict(float), defaultdict(int)
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            n[r["Kernel_Name"]] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


def bench_name(mangled):
    """the kernel naming of bench.py's roofline record for the kernels it can be the dominant one of"""
    m = re.search(r"dcn_window_rows_kernel(?:I|<)(DF16_|f|_Float16|float)[^a-zA-Z]*(Lb1|true)", mangled)
    if m:
        return "dcn_window_rows_kernel<128x64,offset conv fused>"
    m = re.search(r"dcn_window_rows_kernel(?:I|<)(DF16_|f|_Float16|float)", mangled)
    if m:
        return f"dcn_window_kernel<128x64,{'f16' if m.group(1) in ('DF16_', '_Float16') else 'f32'}>"
    m = re.search(r"dcn_window_kernelILi(\d+)ELi\d+ELb[01]E(DF16_|f)", mangled)
    if m:
        return f"dcn_window_kernel<128x{m.group(1)},{'f16' if m.group(2) == 'DF16_' else 'f32'}>"
    m = re.search(r"conv3x3_halo_pair2_kernel(?:ILi|<)(\d+)", mangled)
    if m:
        return f"conv3x3_halo_pair2_kernel<256x{m.group(1)},f16x3>"
    m = re.search(r"conv3x3_halo_pair_kernel(?:ILi|<)(\d+)", mangled)
    if m:
        return f"conv3x3_halo_pair_kernel<256x{m.group(1)},f16x3>"
    if "dcn_split_window_kernel" in mangled:
        return "dcn_f16x3_window_kernel<8x16,64>"
    m = re.search(r"conv3x3_halo_kernel(?:ILi|<)(\d+)", mangled)
    if m:
        o = "f32" if ("float" in mangled or mangled.rstrip().endswith("fEv8ConvArgs")) else "f16"
        return f"conv3x3_halo_kernel<256x{m.group(1)},{o}>"
    if "head_fused_kernel" in mangled:
        return "head_fused_kernel<128x256,f16>"
    if "dla_base_fused_kernel" in mangled:
        return "dla_base_fused_kernel<u8|f32 -> 32ch,f16>"
    if "dec_tile_kernel" in mangled:
        return "dec_tile_kernel"
    return None


def main():
    out, tag = sys.argv[1], sys.argv[2]
    fetch = per_kernel(os.path.join(out, "fetch"), "FETCH_SIZE")
    write = per_kernel(os.path.join(out, "write"), "WRITE_SIZE")
    rows, named = [], {}
    for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
        f, n = fetch[k]
        w = write.get(k, (0.0, 0))[0]
        b = int((2 * f + w) * 1024)
        rows.append((k, n, f, w, b))
        bn = bench_name(k)
        if bn and bn not in named:
            named[bn] = b
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_per_kernel.csv"), "w") as f:
        f.write("kernel,launches,FETCH_SIZE_KB_per_launch_raw,WRITE_SIZE_KB_per_launch,hbm_bytes_per_launch_corrected\n")
        for k, n, fs, ws, b in rows:
            f.write(f"\"{k}\",{n},{fs:.1f},{ws:.1f},{b}\n")
    with open(os.path.join(ROOT, "profiles", f"{tag}_hbm_traffic.json"), "w") as f:
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_hbm.sh) over "
                             "tools/eager_forward.py 64 2; bytes = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024 (gfx950 FETCH_SIZE "
                             "correction, MI355X_MICROARCH.md HBM section); average over the launches of the forwards",
                   "csrc_sha16": csrc_sha16(), "bytes_per_launch": named}, f, indent=1)
    print(json.dumps(named, indent=1))


if __name__ == "__main__":
    main()
