#!/usr/bin/env python3
"""Turn rocprofv3 output directories into the markdown tables kept under profiles/.

    python tools/profile_summary.py stats <dir>            kernel-trace --stats run -> table of kernels
    python tools/profile_summary.py pmc <dir> [<dir> ...]   --pmc runs -> per-kernel average of every counter
    python tools/profile_summary.py gather <dir> <n_atoms> <edge dtype>
        tools/gpu_pmc_gather.sh's passes (<dir>/live.json, trace/, fetch/, write/, tcc/) -> the conv-layer edge kernel's HBM
        traffic per launch, its L2 hit rate and the neighbour-gather figure in counter bytes (<dir>/gather.json + markdown)
    python tools/profile_summary.py gatherjson <out.json> <dir> [<dir> ...]
        merge the gather.json records of several `gather` directories into the record bench.py reports as
        roofline.neighbour_gather_hbm (profiles/gather_hbm.json), stamped with the hash of the kernel sources
    python tools/profile_summary.py pmcjson <fetch_dir> <write_dir> <kernel substring> <out.json> [<trace_dir>]
        (with <trace_dir>, a --kernel-trace --stats run of the same build: the kernel's average launch duration goes into the
        record too, which bench.py reports as roofline.rocprof_avg_launch_ms)
        HBM bytes per launch of one kernel (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md section HBM), stamped with
        the hash of the kernel sources (bench.kernel_source_hash) so that bench.py only reports it for the build it was taken on
"""
import csv
import glob
import importlib.util
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "")


def stats(d):
    rows = []
    for f in find(d, "*kernel_stats.csv"):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
    for r in rows[:24]:
        print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | "
              f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")


def counters(dirs):
    acc = defaultdict(lambda: defaultdict(list))          # kernel -> counter -> per-dispatch sums
    for d in dirs:
        per = defaultdict(float)
        meta = {}
        for f in find(d, "*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                key = (f, r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])
                per[key] += float(r["Counter_Value"])
                meta[(r["Kernel_Name"])] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
        for (f, disp, k, c), v in per.items():
            acc[k][c].append(v)
    return acc


def pmc(dirs):
    acc = counters(dirs)
    for k in sorted(acc, key=lambda k: -sum(len(v) for v in acc[k].values())):
        if not any(s in k for s in ("k_conv", "k_edge_enc", "k_node")):
            continue
        print(f"\n### `{short(k)}`\n\n| counter | dispatches | average per launch |\n|---|---|---|")
        for c in sorted(acc[k]):
            v = acc[k][c]
            v = v[len(v) // 5:] if len(v) >= 5 else v        # drop warm-up launches
            print(f"| {c} | {len(v)} | {sum(v) / len(v):.6g} |")


def pmcjson(fetch_dir, write_dir, kernel, out, trace_dir=None):
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fa, wa = counters([fetch_dir]), counters([write_dir])
    def avg(acc, cname):
        for k in acc:
            if kernel in k and cname in acc[k]:
                v = acc[k][cname]
                v = v[len(v) // 5:] if len(v) >= 5 else v
                return sum(v) / len(v), len(v), k
        raise SystemExit(f"no {cname} for a kernel matching {kernel!r}")
    fs, nf, kname = avg(fa, "FETCH_SIZE")
    ws, nw, _ = avg(wa, "WRITE_SIZE")
    rec = {"kernel": short(kname), "FETCH_SIZE_KB": fs, "WRITE_SIZE_KB": ws, "launches_averaged": [nf, nw],
           "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (each with --kernel-trace only), "
                     "bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary, C2 workload",
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> doubled "
                         "(MI355X_MICROARCH.md section HBM); WRITE_SIZE used as reported (uncalibrated)",
           "hbm_bytes_per_launch": (2.0 * fs + ws) * 1024.0,
           "kernel_source_sha256_16": bench.kernel_source_hash(),
           "algorithmic_bytes_per_launch_note": "e_frag 512 B/edge + pieces ~40 B/edge = ~3.5e8 B at E=6.34e5; S/D/hn gathers are L2-resident"}
    if trace_dir:
        for f in find(trace_dir, "*kernel_stats.csv"):
            for r in csv.DictReader(open(f)):
                if kernel in short(r["Name"]) and "rocprof_avg_launch_us" not in rec:
                    rec["rocprof_avg_launch_us"] = float(r["AverageNs"]) / 1e3
                    rec["rocprof_calls"] = int(r["Calls"])
                    rec["rocprof_source"] = ("rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-secondary "
                                             "--steps 50 --warmup 5 --workload c2 (tools/gpu_profile_round.sh, same kernel sources)")
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


def gather(d, n_atoms, dtype):
    n_atoms = int(n_atoms)
    live = json.load(open(os.path.join(d, "live.json")))[0]
    E = live["edges"]
    kern = {"f32": "k_conv_edge<", "bf16": "k_conv_edge_bf16", "f16x3": "k_conv_edge_f16x3"}[dtype]
    trace_us = None
    for f in find(os.path.join(d, "trace"), "*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if kern in short(r["Name"]):
                trace_us = float(r["AverageNs"]) / 1e3
    def avg(sub, cname):
        acc = counters([os.path.join(d, sub)])
        for k in acc:
            if kern in short(k) and cname in acc[k]:
                v = acc[k][cname]
                v = v[len(v) // 5:] if len(v) >= 5 else v
                return sum(v) / len(v)
        return None
    fetch_kb, write_kb = avg("fetch", "FETCH_SIZE"), avg("write", "WRITE_SIZE")
    hit, miss, req = avg("tcc", "TCC_HIT_sum"), avg("tcc", "TCC_MISS_sum"), avg("tcc", "TCC_REQ_sum")
    e_bytes = {"f32": 512.0, "bf16": 256.0, "f16x3": 512.0}[dtype]
    alg_gather = E * 1028.0 + n_atoms * 1024.0                     # SURVEY.md 8d, per layer: idx + hn[src] + S[src] rows, D rows, agg
    row_bytes = 256.0 if dtype == "bf16" else 512.0                        # bf16 edge MLP: fp16 node tables (round 6)
    mandatory = E * (e_bytes + 4.0 + 4.0 + 36.0) + n_atoms * 3 * row_bytes  # e stream + col / erow + pieces; each node-table row ONCE
    t_live = live["conv_ms_per_launch"] * 1e-3
    hbm = (2.0 * fetch_kb + write_kb) * 1024.0 if fetch_kb is not None and write_kb is not None else None
    rec = {"n_atoms": n_atoms, "edges": E, "edge_dtype": dtype, "kernel": kern.rstrip("<"),
           "conv_ms_per_launch_live": live["conv_ms_per_launch"], "conv_us_per_launch_rocprofv3": trace_us,
           "ms_per_step": live["ms_per_step"], "atom_steps_per_s": live["atom_steps_per_s"],
           "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "TCC_HIT": hit, "TCC_MISS": miss, "TCC_REQ": req,
           "l2_hit_rate": hit / (hit + miss) if hit is not None and miss else None,
           "hbm_bytes_per_launch": hbm,
           "hbm_GB_per_s": hbm / t_live / 1e9 if hbm else None,
           "hbm_frac_of_8TBs": hbm / t_live / 8e12 if hbm else None,
           "algorithmic_gather_bytes_per_launch": alg_gather,
           "algorithmic_gather_frac_of_8TBs": alg_gather / t_live / 8e12,
           "mandatory_bytes_per_launch": mandatory,
           "hbm_over_mandatory": hbm / mandatory if hbm else None,
           "correction": "FETCH_SIZE x 2 (gfx950 counts 64 B per 128-B request, MI355X_MICROARCH.md section HBM); WRITE_SIZE as "
                         "reported; Infinity-Cache hits are counted by both (fabric-side request counters)",
           "node_tables_MB": n_atoms * 3 * 512.0 / 1e6}
    json.dump(rec, open(os.path.join(d, "gather.json"), "w"), indent=1)
    g = lambda v, f="%.4g": "n/a" if v is None else f % v
    print(f"### {n_atoms} atoms, {E} edges, edge dtype {dtype}: `{rec['kernel']}`\n")
    print("| quantity | value |\n|---|---|")
    print(f"| step | {live['ms_per_step']:.3f} ms, {live['atom_steps_per_s']:.3e} atom-steps/s |")
    print(f"| conv-layer edge kernel | {live['conv_ms_per_launch'] * 1e3:.1f} us per launch (live HIP events), {g(trace_us, '%.1f')} us (rocprofv3 trace) |")
    print(f"| node tables hn + S + D | {rec['node_tables_MB']:.0f} MB (L2 8 x 4 MiB, Infinity Cache 256 MiB) |")
    print(f"| FETCH_SIZE x 2 | {g(None if fetch_kb is None else 2 * fetch_kb * 1024 / 1e6)} MB per launch |")
    print(f"| WRITE_SIZE | {g(None if write_kb is None else write_kb * 1024 / 1e6)} MB per launch |")
    print(f"| memory-side traffic | {g(None if hbm is None else hbm / 1e6)} MB per launch = {g(rec['hbm_GB_per_s'])} GB/s = **{g(rec['hbm_frac_of_8TBs'], '%.3f')}** of 8 TB/s |")
    print(f"| mandatory bytes (e stream, indices, pieces, every node-table row once) | {mandatory / 1e6:.4g} MB per launch; traffic / mandatory = {g(rec['hbm_over_mandatory'], '%.2f')} |")
    print(f"| L2 (TCC) | hit {g(hit)} miss {g(miss)} req {g(req)} per launch: hit rate {g(rec['l2_hit_rate'], '%.3f')} |")
    print(f"| SURVEY 8d gather figure (algorithmic: E x 1 028 B + N x 1 024 B) | {alg_gather / 1e6:.4g} MB per launch = {alg_gather / t_live / 1e9:.4g} GB/s = {rec['algorithmic_gather_frac_of_8TBs']:.3f} of 8 TB/s |")


def gatherjson(out, dirs):
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    keep = ("n_atoms", "edges", "kernel", "conv_ms_per_launch_live", "conv_us_per_launch_rocprofv3", "hbm_bytes_per_launch",
            "hbm_GB_per_s", "hbm_frac_of_8TBs", "hbm_over_mandatory", "l2_hit_rate", "algorithmic_gather_frac_of_8TBs",
            "FETCH_SIZE_KB", "WRITE_SIZE_KB", "atom_steps_per_s")
    rec = {"records": {}}
    for d in dirs:
        r = json.load(open(os.path.join(d, "gather.json")))
        rec["records"][f"{r['edge_dtype']}_{r['n_atoms']}"] = {k: r[k] for k in keep}
    rec["source"] = ("tools/gpu_pmc_gather.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum in separate passes "
                     "(each with --kernel-trace only) on the LJ workload at 10^5 / 10^6 atoms (node tables beyond L2 / Infinity Cache); "
                     "FETCH_SIZE x 2 (gfx950); fabric-side counters, Infinity-Cache hits included; profiles/r05_gather_hbm.md")
    rec["kernel_source_sha256_16"] = bench.gather_source_hash()
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "stats":
        stats(sys.argv[2])
    elif cmd == "pmc":
        pmc(sys.argv[2:])
    elif cmd == "gather":
        gather(*sys.argv[2:5])
    elif cmd == "gatherjson":
        gatherjson(sys.argv[2], sys.argv[3:])
    elif cmd == "pmcjson":
        pmcjson(*sys.argv[2:7])
