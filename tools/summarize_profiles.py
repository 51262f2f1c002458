#!/usr/bin/env python
"""Turns gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the committed summaries under profiles/:
   profiles/<tag>_kernel_stats.md   per-kernel time table (rocprofv3 --kernel-trace --stats)
   profiles/<tag>_pmc.md            PMC counters of the hot kernels
   profiles/traffic_<cfg>.json      HBM bytes per launch per stage (FETCH_SIZE / WRITE_SIZE passes), read by bench.py
   profiles/alu_<cfg>.json          matrix / vector pipe busy fractions per stage (SQ passes), read by bench.py
Both JSON files carry `_stamp` = mi_rast_version() of the library the passes ran with (a hash of sources, headers and flags);
bench.py ignores them when another build is loaded.   python tools/summarize_profiles.py <tag> [cfg3|cfg2|cfg5]
"""
import collections
import csv
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
sfx = "" if cfg == "cfg3" else f"_{cfg}"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}{sfx}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    if "onesweep_iteration" in n: return "rocprim::radix_sort_onesweep_iteration (depth sort of P keys)"
    if "onesweep_global" in n or "onesweep_hist" in n: return "rocprim::radix_sort_onesweep_histograms"
    if "rocprim" in n: return "rocprim::" + n.split("::")[-1][:50]
    return n[:90]


# kernel -> stage of bench.py's roofline.stages.  Matched on the kernel's BASE name, plus its first template argument where one
# kernel serves two stages (bin_spans_kernel<EMIT, ...>): round 5 keyed this table on full instance names, the instances grew a second
# argument, and the emit / count kernels silently dropped out of traffic_*.json.  A kernel of the library that holds more than 1 % of
# the trace and maps to no stage is an ERROR now (stage_of(..., strict=True)).
STAGE_BY_BASE = {"blend_bwd_wave_kernel": "blend_bwd", "blend_bwd_kernel": "blend_bwd",
                 "blend_fwd_kernel": "blend_fwd", "blend_fwd_x3_kernel": "blend_fwd", "blend_fwd_wave_kernel": "blend_fwd", "blend_fwd_wave_rgb_kernel": "blend_fwd",
                 "run_bounds_from_walks_kernel": "blend_fwd",
                 "preprocess_fwd_kernel": "preprocess",
                 "tile_sort_kernel": "tile_sort", "tile_sort_wave_kernel": "tile_sort", "reuse_image_state_kernel": "tile_scan", "tile_ranges_kernel": "tile_scan",
                 "geometry_bwd_kernel": "geom_bwd", "unpack_mask_kernel": "geom_bwd"}
STAGE_BY_FIRST_ARG = {"bin_spans_kernel": {"true": "emit", "false": "tile_scan"}}
# kernels of the FULL-list mode (parity tests; bench.py runs them once, in its counter step): listed in the tables, part of no stage
PARITY_ONLY = ("bin_count_kernel", "bin_ranks_kernel", "scan_partials_kernel", "verify_entries_kernel")


def base_and_args(name):
    """'void mirast::bin_spans_kernel<true, 1024>(int, ...)' -> ('bin_spans_kernel', ['true', '1024'], True)"""
    n = re.sub(r"\(.*", "", name).replace("void ", "").strip()
    ours = n.startswith("mirast::")
    n = n.replace("mirast::", "")
    m = re.match(r"([A-Za-z_0-9:]+)(?:<(.*)>)?$", n)
    if not m:
        return n, [], ours
    return m.group(1), [a.strip() for a in (m.group(2) or "").split(",") if a.strip()], ours


def stage_of(name):
    base, args, _ = base_and_args(name)
    if base in STAGE_BY_FIRST_ARG:
        return STAGE_BY_FIRST_ARG[base].get(args[0] if args else "")
    return STAGE_BY_BASE.get(base)


def pmc_key(name):
    """Row key of the PMC tables: the base name; the first template argument kept where it selects the stage."""
    base, args, _ = base_and_args(name)
    return f"{base}<{args[0]}>" if base in STAGE_BY_FIRST_ARG and args else base


rows = list(csv.DictReader(open(os.path.join(src, "trace_kernel_stats.csv"))))
agg = {}
for r in rows:
    a = agg.setdefault(short(r["Name"]), [0, 0]); a[0] += int(r["Calls"]); a[1] += int(r["TotalDurationNs"])
tot = sum(v[1] for v in agg.values())
unmapped = {}
for r in rows:
    base, args, ours = base_and_args(r["Name"])
    if ours and stage_of(r["Name"]) is None and not base.startswith(("knn", "contrastive", "fingerprint")) and base not in PARITY_ONLY:
        unmapped[base] = unmapped.get(base, 0) + int(r["TotalDurationNs"])
bad = {k: v for k, v in unmapped.items() if v > 0.01 * tot}
if bad:
    sys.exit(f"summarize_profiles: kernels with > 1 % of the trace map to no stage of bench.py: {bad} -- extend STAGE_BY_BASE / STAGE_BY_FIRST_ARG")
bench_line = open(os.path.join(src, "bench_line.json")).read().strip().splitlines()[-1]
stamp = (json.loads(bench_line).get("roofline") or {}).get("library")
with open(os.path.join(dst, f"{tag}_kernel_stats{sfx}.md"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats ({tag})\n\n")
    f.write(f"Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --config {cfg} --steps 10 --warmup 2 --settle 0 "
            f"--dist-blocks 0 --no-cpu-baseline` on 1x MI355X (BASELINE {cfg}); library `{stamp}`. Kernel names shortened; template "
            "instances of one kernel merged. Each bench step launches every kernel once; calls also include the warm-up, "
            "the per-stage timing steps and one counter step.\n\n")
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f"| `{k}` | {c} | {t/1e6:.3f} | {t/c/1e3:.1f} | {100*t/tot:.2f} |\n")
    f.write(f"\nUn-profiled bench line of the same build (`python bench.py --config {cfg} --steps 20 --warmup 3`), printed on the box right after the "
            "passes above -- i.e. BEFORE this build's PMC summaries existed, hence `traffic` / `alu` null and `pmc: ... ignored` in it; the lines "
            f"with the summaries in place are in profiles/{tag}_bench_lines.md:\n\n```json\n" + bench_line + "\n```\n")


def pmc(name):
    path = os.path.join(src, f"{name}_counter_collection.csv")
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return out
    for r in csv.DictReader(open(path)):
        out[pmc_key(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


fetch, write = pmc("fetch"), pmc("write")
traffic = {}
untracked = {}   # kernels of the PMC passes that belong to no stage (torch fills, knn, ...): listed, not summed
with open(os.path.join(dst, f"{tag}_pmc{sfx}.md"), "w") as f:
    f.write(f"# rocprofv3 PMC passes ({tag}, {cfg}, library {stamp})\n\nSeparate passes per counter group (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, two SQ groups), "
            "`--kernel-trace` only, same bench command. FETCH_SIZE / WRITE_SIZE are in KiB per dispatch (averaged over the "
            "dispatches of a kernel). Per `/opt/skills/guides/MI355X_MICROARCH.md` (HBM): on gfx950 FETCH_SIZE reports exactly "
            "half of the bytes of a wide coalesced stream (it tallies 128-B requests at 64 B), so `traffic = (2*FETCH_SIZE + "
            "WRITE_SIZE) * 1024`; the uncorrected sum is listed too. Infinity-Cache hits are counted, not excluded.\n\n")
    f.write("| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | traffic corrected MB | uncorrected MB |\n|---|---|---|---|---|\n")
    for k in sorted(set(fetch) | set(write)):
        fs = fetch[k].get("FETCH_SIZE", [0]); ws = write[k].get("WRITE_SIZE", [0])
        fa, wa = sum(fs) / len(fs), sum(ws) / len(ws)
        corr, raw = (2 * fa + wa) * 1024, (fa + wa) * 1024
        f.write(f"| `{k}` | {fa:.0f} | {wa:.0f} | {corr/1e6:.1f} | {raw/1e6:.1f} |\n")
        st = stage_of(k)
        if st:
            t = traffic.setdefault(st, {"bytes_per_launch": 0.0, "kernels": []})
            t["bytes_per_launch"] += corr; t["kernels"].append(k)
        elif corr > 0:
            untracked[k] = corr
    for name in ("sq1", "sq2"):
        p = pmc(name)
        if not p: continue
        f.write(f"\n## SQ counters ({name}); per dispatch averages, summed over the chip as rocprofv3 reports them\n\n")
        for k, cs in p.items():
            f.write(f"**{k}**\n\n| counter | value |\n|---|---|\n")
            for c, v in sorted(cs.items()):
                f.write(f"| {c} | {sum(v)/len(v):.0f} |\n")
            f.write("\n")
# matrix / vector pipe busy fractions of the blend kernels (bench.py: roofline.alu).  SIMD-cycles of a dispatch =
# GRBM_GUI_ACTIVE (summed over the 8 XCDs by rocprofv3) / 8 x 1024 SIMDs; SQ_VALU_MFMA_BUSY_CYCLES is in cycles,
# SQ_ACTIVE_INST_VALU in quad-cycles.
alu = {}
sq1, sq2 = pmc("sq1"), pmc("sq2")
avg = lambda d, k: (sum(d[k]) / len(d[k])) if k in d and d[k] else None
for k in set(sq1) & set(sq2):
    st = stage_of(k)
    gui, mf, va = avg(sq2[k], "GRBM_GUI_ACTIVE"), avg(sq2[k], "SQ_VALU_MFMA_BUSY_CYCLES"), avg(sq1[k], "SQ_ACTIVE_INST_VALU")
    if not st or not gui:
        continue
    simd_cycles = gui / 8.0 * 1024.0
    alu[st] = {"kernel": k, "mfma_busy_frac": round((mf or 0) / simd_cycles, 4), "valu_busy_frac": round(4.0 * (va or 0) / simd_cycles, 4),
               "lds_bank_conflict_frac": round((avg(sq2[k], "SQ_LDS_BANK_CONFLICT") or 0) / max(1.0, avg(sq2[k], "SQ_LDS_IDX_ACTIVE") or 1.0), 4),
               "source": f"profiles/{tag}_pmc{sfx}.md"}
alu["_stamp"] = stamp
alu["_source"] = f"profiles/{tag}_pmc{sfx}.md"
json.dump(alu, open(os.path.join(dst, f"alu_{cfg}.json"), "w"), indent=1)
for st in traffic.values():
    st["bytes_per_launch"] = round(st["bytes_per_launch"])
traffic["_sum_of_stages"] = sum(st["bytes_per_launch"] for st in traffic.values())   # = the sum over every mapped kernel of the PMC table above
traffic["_untracked_kernels"] = {k: round(v) for k, v in untracked.items()}
traffic["_source"] = f"profiles/{tag}_pmc{sfx}.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per the gfx950 note)"
traffic["_stamp"] = stamp
json.dump(traffic, open(os.path.join(dst, f"traffic_{cfg}.json"), "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_kernel_stats{sfx}.md")).read()[:3000])
print(json.dumps(traffic, indent=1))
