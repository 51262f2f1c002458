"""gpurun_out/lines_<tag>/*.log (tools/final_lines.sh) -> profiles/<tag>_bench_lines.md:  python tools/bench_lines_md.py r03"""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"lines_{tag}")
runs = [("cfg3_default", "python bench.py"), ("cfg3", "python bench.py --ref-on-gpu --steps 30"), ("cfg2", "python bench.py --config cfg2"),
        ("cfg3s", "python bench.py --config cfg3s"), ("cfg5", "python bench.py --config cfg5 --ref-on-gpu"), ("cfg1", "python bench.py --config cfg1"),
        ("dist", "python bench.py --dist-single --steps 10 --warmup 3"), ("dist_rsag", "python bench.py --dist-single --rs-ag --steps 10 --warmup 3"), ("fastexp", "python bench.py --fast-exp --no-cpu-baseline"),
        ("frozen", "python bench.py --frozen-geometry --features-only-grad --views 16 --no-cpu-baseline"),
        ("frozen_cfg5", "python bench.py --config cfg5 --frozen-geometry --features-only-grad --views 8 --no-cpu-baseline")]
out = [f"# Bench lines of the final round build ({tag})",
       "`python bench.py [--config ...]` on 1x MI355X through gpurun (default: 100 timed steps after 10 warm-up steps and the settling blocks); "
       "`--ref-on-gpu` adds `reference_on_gpu` (oracle/_ref = the reference's own kernels, hipify-perl at build time, timed on the same workload "
       "and GPU after the timed region).  `timing` = median / p10 / p90 over 20 untimed blocks of ten steps; `roofline.traffic` / `alu` are shown "
       "only when the committed PMC summary carries the stamp of the library being timed (`roofline.library`).", ""]
frozen_rows = []
fo_rows = []
summary = ["| run | views/s | sustained views/s (>= 2 s back to back) | ms/step | median (p10..p90) | dominant kernel: frac | whole view: frac (replaced stages left out) / by SURVEY bytes / by traffic |", "|---|---|---|---|---|---|---|"]
for name, cmd in runs:
    path = os.path.join(src, name + ".log")
    if not os.path.exists(path):
        continue
    lines = open(path).read().strip().splitlines()
    js = next((l for l in reversed(lines) if l.startswith("{")), None)
    note = [l for l in lines if l.startswith("[bench]")]
    out += [f"## `{cmd}`", ""]
    if note:
        out += ["```", note[0], "```", ""]
    if js:
        d = json.loads(js)
        r = d.get("roofline") or {}
        wv = r.get("whole_view") or {}
        t = d.get("timing") or {}
        summary.append(f"| `{cmd}` | {d['value']} | {(d.get('sustained') or {}).get('value')} | {d['ms_per_step']} | {t.get('ms_per_step_median')} ({t.get('ms_per_step_p10')}..{t.get('ms_per_step_p90')}) | "
                       f"{r.get('kernel')}: {r.get('frac')} | {wv.get('frac')} / {wv.get('frac_survey_bytes')} / {wv.get('frac_traffic')} |")
        fg = d.get("frozen_geometry")
        if fg:
            frozen_rows.append(f"| `{cmd}` | {d['value']} | {fg['views_per_s']} | {fg['views_per_s_all_hits']} | {fg['views_per_s_without_cache_same_sequence']} | {fg['hit_rate']} | {fg['bytes_cached_per_view']} |")
        fo = d.get("features_only_backward")
        if fo:
            st = fo.get("stages_ms") or {}
            fo_rows.append(f"| `{cmd}` | {d['value']} | {fo['views_per_s_default_backward_same_loop']} | {fo['views_per_s']} | {st.get('blend_bwd')} | "
                           f"{fo.get('views_per_s_with_frozen_geometry_all_hits')} | {fo['dL_dfeatures_max_abs_diff_vs_default']} of {fo['dL_dfeatures_max_abs']} |")
        out += ["```json", js, "```", ""]
if frozen_rows:
    summary += ["", "Frozen-geometry reuse (opt-in, DESIGN section 12; a separately labelled figure, never `value`): the same fwd+bwd steps cycling over the "
                "given number of cameras, first visits inside the timed region.", "",
                "| run | headline views/s (everything recomputed) | frozen_geometry views/s (first visits included) | all hits | same camera sequence without the cache | hit rate | bytes cached per view |",
                "|---|---|---|---|---|---|---|"] + frozen_rows
if fo_rows:
    summary += ["", "Features-only backward (opt-in / automatic when autograd asks for the colour gradient alone, DESIGN section 13; a separately labelled "
                "figure, never `value`): the same fwd+bwd steps, the backward computing dL_dcolors_precomp only.", "",
                "| run | headline views/s (all eight gradients) | the same through the figure's loop | features_only_backward views/s | its blend_bwd ms | "
                "with the geometry cache too, all hits | max abs difference of dL_dfeatures vs the default backward |",
                "|---|---|---|---|---|---|---|"] + fo_rows
out[3:3] = summary + [""]
open(os.path.join(root, "profiles", f"{tag}_bench_lines.md"), "w").write("\n".join(out) + "\n")
print("\n".join(summary))
