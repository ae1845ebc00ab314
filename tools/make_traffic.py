"""GPU box: regenerate profiles/rNN/traffic.json from PMC passes of THE LIBRARY THAT IS IN THE TREE, and stamp it with that
library's build string (bags_build_info(): hash of the kernel sources + last commit that touched them).  bench.py copies
roofline.traffic from the newest traffic.json whose stamp matches the library it is timing, and prints null otherwise -- the
round-4 file was assembled by hand from an earlier build's counters and went stale when the emission changed.

Counters are collected as MI355X_MICROARCH.md prescribes: separate `rocprofv3 --pmc` runs (no trace domains beside them), the
program itself behind `--` (python3 bench.py ...), FETCH_SIZE / WRITE_SIZE in KiB -> bytes x 1024; bench.py reports
traffic = 2 x fetch + write (the guide's gfx950 correction for wide reads; an upper bound for the 64-byte line gathers of
the blend kernels) and fetch + write as `traffic_uncorrected`.

usage: python3 tools/make_traffic.py [--out gpurun_out/traffic.json] [--tile-bounds opacity] [--sm 0.5]
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bundle-adjusting-gaussian-splatting_amd")]

# kernel-name substring -> stage name of bench.py's stage_ms / alg table
KERNELS = (("blend_bwd_scan_kernel", "blend_bwd"), ("blend_fwd_rows_kernel", "blend_fwd"), ("preprocess_fwd", "preprocess_fwd"),
           ("preprocess_bwd_kernel", "preprocess_bwd"), ("emit_binned_kernel", "emit_binned"), ("tile_prefix_kernel", "tile_prefix"),
           ("pose_reduce_kernel", "pose_reduce"))
PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"),
          ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES"),
          ("SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
          ("SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL"))


def pmc_pass(counters, bench_args):
    out = tempfile.mkdtemp(prefix="pmc_", dir=os.path.join(ROOT, "gpurun_out"))
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", out, "--", "python3", os.path.join(ROOT, "bench.py"),
           "--no-cpu-baseline", "--no-profile", "--no-aabb-leg", "--no-v4-leg", "--no-lazy-leg", "--no-median-leg", "--steps", "3",
           "--warmup", "1", "--settle-steps", "20", *bench_args]
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
    files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if files:
        for row in csv.DictReader(open(files[0])):
            for pat, stage in KERNELS:
                if pat in row["Kernel_Name"]:
                    acc[stage][row["Counter_Name"]].append(float(row["Counter_Value"]))
                    break
    else:
        print("no counter file for", counters, r.stderr[-800:], file=sys.stderr)
    shutil.rmtree(out, ignore_errors=True)
    return {st: {c: sum(v) / len(v) for c, v in d.items()} for st, d in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "traffic.json"))
    ap.add_argument("--tile-bounds", default="opacity")
    ap.add_argument("--sm", type=float, default=0.5)
    ap.add_argument("--P", type=int, default=500000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    args = ap.parse_args()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    from bags_raster import _lib
    build = _lib.load().bags_build_info().decode()
    bench_args = ["--tile-bounds", args.tile_bounds, "--sm", str(args.sm), "--P", str(args.P), "--width", str(args.width), "--height", str(args.height)]
    merged = collections.defaultdict(dict)
    for counters in PASSES:
        got = pmc_pass(counters, bench_args)
        for st, d in got.items():
            merged[st].update(d)
        print(counters, {st: {c: round(v) for c, v in d.items()} for st, d in got.items() if st.startswith("blend")}, flush=True)
    out = {"_comment": "HBM traffic and SQ counters per launch (means over the launches of a short bench run), rocprofv3 --pmc, one pass per "
                       "counter group; FETCH_SIZE / WRITE_SIZE are KiB counters x 1024.  bench.py: traffic = 2 x fetch_bytes + write_bytes "
                       "(MI355X_MICROARCH.md, gfx950), traffic_uncorrected = fetch_bytes + write_bytes.  Written by tools/make_traffic.py.",
           "key": {"P": args.P, "width": args.width, "height": args.height, "sm": args.sm, "tile_bounds": args.tile_bounds},
           "build": build, "source": "tools/make_traffic.py on " + build}
    names = {"SQ_INSTS_VALU": "valu_insts", "SQ_INSTS_SALU": "salu_insts", "SQ_INSTS_LDS": "lds_insts", "SQ_WAVES": "waves",
             "SQ_ACTIVE_INST_VALU": "active_inst_valu_quads", "SQ_ACTIVE_INST_LDS": "active_inst_lds_quads",
             "SQ_BUSY_CU_CYCLES": "busy_cu_cycles", "SQ_WAVE_CYCLES": "wave_cycles_quads", "SQ_LDS_BANK_CONFLICT": "lds_bank_conflict_cycles",
             "SQ_LDS_IDX_ACTIVE": "lds_idx_active_cycles", "SQ_WAIT_INST_LDS": "wait_inst_lds_quads", "SQ_WAIT_INST_ANY": "wait_inst_any_quads",
             "SQ_LDS_ADDR_CONFLICT": "lds_addr_conflict_cycles", "SQ_LDS_UNALIGNED_STALL": "lds_unaligned_stall_cycles"}
    for st, d in merged.items():
        e = {}
        if "FETCH_SIZE" in d:
            e["fetch_bytes"] = int(d["FETCH_SIZE"] * 1024)
        if "WRITE_SIZE" in d:
            e["write_bytes"] = int(d["WRITE_SIZE"] * 1024)
        for c, n in names.items():
            if c in d:
                e[n] = int(d[c])
        out[st] = e
    json.dump(out, open(args.out, "w"), indent=1)
    print("wrote", args.out, "for", build)


if __name__ == "__main__":
    main()
