"""Per-kernel averages from rocprofv3 counter-collection CSVs (one or more --pmc passes).

  python tools/pmc_summary.py sq   <counter_collection.csv>... > profiles/rNN_sq_counters_per_kernel.csv
  python tools/pmc_summary.py hbm  <counter_collection.csv>... > profiles/rNN_hbm_traffic_per_kernel.csv
  python tools/pmc_summary.py json <counter_collection.csv>... > profiles/rNN_traffic.json

`hbm`/`json`: HBM bytes per launch = 2 x FETCH_SIZE (KB, gfx950 counts 128-B requests at 64 B:
/opt/skills/guides/MI355X_MICROARCH.md) + WRITE_SIZE (KB).  `json` keeps the keys bench.py reads."""
import collections
import csv
import json
import re
import sys

BENCH_KEYS = {  # bench.py kernel key -> kernel-name prefixes whose launches it covers
    "circuit_jets_bwd": ["k_jets_bwd<"], "circuit_jets_fwd": ["k_jets_fwd<"], "pre_bwd": ["k_pre_bwd<4, 6>"],
    "pre_fwd": ["k_pre_fwd<4, 6>"], "post": ["k_post<4, 6, 2>", "k_post_wg<4, 6>", "k_post_fused<4, 6>"],
    "stage_circuit_bwd": ["k_circ_bwd_both<", "k_circ_bwd_both2<"], "stage_circuit_fwd": ["k_circ_fwd_both<"],
    "fold_rows": ["k_fold_rows"], "adam": ["k_adam_fast<true>"],
    "stage_pre_fwd": ["k_pre_fwd_both<"], "stage_pre_bwd": ["k_pre_bwd_both<"],
    "stage_post": ["k_post_both<", "k_post_wg_both<", "k_post_fused_both<"],
}


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)[:60]


def load(paths):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> per-dispatch values
    for p in paths:
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(p)):
            key = (p, r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            names[(p, r["Dispatch_Id"])] = short(r["Kernel_Name"])
        for (pp, did, cn), v in per_dispatch.items():
            acc[names[(pp, did)]][cn].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    mode, paths = sys.argv[1], sys.argv[2:]
    avg = load(paths)
    if mode == "sq":
        print("# rocprofv3 --pmc SQ_* passes, averages per launch")
        print("kernel,VALU_insts_per_wave,SALU_insts_per_wave,wave_cycles_per_wave,wait_any_frac,counters")
        for k, c in sorted(avg.items()):
            w = c.get("SQ_WAVES", 0) or 1
            cyc = c.get("SQ_WAVE_CYCLES", 0)
            print(f"{k},{c.get('SQ_INSTS_VALU', 0) / w:.0f},{c.get('SQ_INSTS_SALU', 0) / w:.0f},{cyc / w:.0f},"
                  f"{(c.get('SQ_WAIT_ANY', 0) / cyc if cyc else 0):.2f},"
                  + " ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())))
    else:
        rows = {}
        for k, c in avg.items():
            f, w = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
            rows[k] = (f, w, int((2 * f + w) * 1024))
        if mode == "hbm":
            print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), averages per launch")
            print("kernel,FETCH_SIZE_KB_raw,WRITE_SIZE_KB,hbm_bytes_fetch_x2")
            for k, (f, w, b) in sorted(rows.items()):
                print(f"{k},{f:.1f},{w:.1f},{b}")
        else:
            out = {}
            for key, subs in BENCH_KEYS.items():
                tot = sum(b for k, (_, _, b) in rows.items() if any(k.startswith(sub) for sub in subs))
                if tot:
                    out[key] = tot
            print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
