"""Run rocprofv3 counter passes on a command and print per-kernel averages.

    python scripts/pmc.py --out gpurun_out/pmc_x --match wmsa -- python3 scripts/bench_wmsa.py

Counters go in separate passes (SQ: 8 slots, TCC: 4 with FETCH_SIZE = 3 and WRITE_SIZE = 2, see
/opt/skills/guides/MI355X_MICROARCH.md), always with --kernel-trace only.  The program after `--` is started
directly by rocprofv3 (no shell / env hop).  Prints, per kernel whose name contains --match: launches, mean
duration and the mean of every counter, plus the derived MFMA-busy fraction
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz)."""
import argparse, collections, csv, glob, os, re, subprocess, sys

PASSES = {
    "sq1": "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE",
    "sq2": "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE",
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
    "tcc": "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum",
    "tcp": "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--match", default="")
    ap.add_argument("--passes", default="sq1,fetch,write")
    ap.add_argument("--family", action="append", default=[], help="name=regex: also print launch-weighted means over all "
                    "kernels whose name matches (e.g. family='gemm_kernel|gemm_glds_kernel|gemm8_kernel')")
    ap.add_argument("--json", default="", help="write the family aggregates to this file")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    os.makedirs(a.out, exist_ok=True)
    dur = collections.defaultdict(list)
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    fams = [f.split("=", 1) for f in a.family]
    fam_ctr = {n: collections.defaultdict(list) for n, _ in fams}
    fam_dur = {n: [] for n, _ in fams}
    for name in a.passes.split(","):
        d = os.path.join(a.out, name)
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *PASSES[name].split(), "--output-format", "csv",
                            "-d", d, "--"] + cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(f"pass {name} failed rc={r.returncode}\n{r.stderr[-2000:]}")
            continue
        disp = {}
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                disp[row["Dispatch_Id"]] = (row["Kernel_Name"], (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        seen = set()
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                kn = row["Kernel_Name"]
                if a.match and a.match not in kn:
                    continue
                key = (re.sub(r"_ZN12_GLOBAL__N_1\d+", "", kn)[:70], int(row["Grid_Size"]) if "Grid_Size" in row else 0)
                ctr[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                did = row["Dispatch_Id"]
                for fn, rx in fams:
                    if re.search(rx, kn):
                        fam_ctr[fn][row["Counter_Name"]].append(float(row["Counter_Value"]))
                        if did in disp and (name, fn, did) not in seen:
                            seen.add((name, fn, did))
                            fam_dur[fn].append(disp[did][1])
                if did in disp and (name, did) not in seen:
                    seen.add((name, did))
                    dur[key].append(disp[did][1])
    for key in sorted(ctr):
        ds = dur.get(key, [0.0])
        # drop warm-up outliers: median
        ds_sorted = sorted(ds)
        med = ds_sorted[len(ds_sorted) // 2]
        print(f"== {key[0]} grid={key[1]}: {len(ds)} profiled launches, median {med:.1f} us")
        for c, vs in sorted(ctr[key].items()):
            print(f"   {c:36s} {sum(vs) / len(vs):16.1f}")
        mf = ctr[key].get("SQ_VALU_MFMA_BUSY_CYCLES")
        if mf and med > 0:
            print(f"   MFMA-busy fraction = {sum(mf) / len(mf) / (1024 * med * 1e-6 * 2.4e9):.4f}")
    out = {}
    for fn, rx in fams:
        c = fam_ctr[fn]
        n = max((len(v) for v in c.values()), default=0)
        if not n:
            continue
        mean = {k: sum(v) / len(v) for k, v in c.items()}
        ds = fam_dur[fn]
        avg_us = sum(ds) / len(ds) if ds else 0.0
        # MI355X_MICROARCH.md, HBM: FETCH_SIZE counts 64 B per 128-B request of wide streaming reads on gfx950 -> x2;
        # WRITE_SIZE is exact for 16-B-per-lane streaming stores and float atomics.  rocprofv3 reports both in KiB.
        fetch = mean.get("FETCH_SIZE"); write = mean.get("WRITE_SIZE")
        traffic = None
        if fetch is not None and write is not None:
            traffic = int(2 * fetch * 1024 + write * 1024)
        out[fn] = {"regex": rx, "profiled_launches": n, "avg_launch_us_profiled": round(avg_us, 2),
                   "counters_mean_per_launch": {k: round(v, 1) for k, v in mean.items()},
                   "traffic_bytes_per_launch": traffic,
                   "traffic_formula": "2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes), gfx950 correction of MI355X_MICROARCH.md"}
        print(f"== FAMILY {fn} ({rx}): {n} launches, mean {avg_us:.1f} us, " + ", ".join(f"{k}={v:.1f}" for k, v in mean.items())
              + (f", traffic {traffic / 1e6:.1f} MB/launch" if traffic else ""))
    if a.json and out:
        import json
        with open(a.json, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
