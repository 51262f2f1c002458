"""Per-basic-block instruction mix of one kernel of csrc/mi_rast.hip as hipcc compiles it for gfx950 (no GPU needed).

   python tools/isa_blocks.py <kernel substring> [<template substring>] [--min N] [--dump LABEL] [-D...]

Prints one line per basic block with >= N instructions: VALU / transcendental / SALU / MFMA / LDS / global loads /
stores / atomics / s_waitcnt, and marks blocks that are targets of a backward branch (loop heads).  --dump LABEL prints
the block's instructions.  Used to count what the blend kernels' inner loops issue per chunk."""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:]]
minn, dump, subs, defs = 20, None, [], []
i = 0
while i < len(args):
    a = args[i]
    if a == "--min":
        minn = int(args[i + 1]); i += 2; continue
    if a == "--dump":
        dump = args[i + 1]; i += 2; continue
    if a.startswith("-D"):
        defs.append(a); i += 1; continue
    subs.append(a); i += 1
tag = "_".join(d[2:] for d in defs) or "base"
out_s = f"/tmp/isa_blocks_{tag}.s"
src = os.path.join(root, "seganygaussians_amd/csrc/mi_rast.hip")
deps = [os.path.join(root, "seganygaussians_amd/csrc", f) for f in os.listdir(os.path.join(root, "seganygaussians_amd/csrc"))]
if not os.path.exists(out_s) or any(os.path.getmtime(d) > os.path.getmtime(out_s) for d in deps):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics",
           "-fno-slp-vectorize", "--cuda-device-only", "-S", src, "-o", out_s] + defs
    subprocess.check_call(cmd)
lines = open(out_s).read().splitlines()
start = None
for n, l in enumerate(lines):
    if l.startswith("_Z") and re.match(r"^\S+:", l) and all(s in l.split(":")[0] for s in subs):
        start = n
        break
if start is None:
    sys.exit("kernel not found: " + " ".join(subs))
end = next(n for n in range(start, len(lines)) if lines[n].startswith(".Lfunc_end"))
print(lines[start][:160])
blocks, cur = [], ["entry", []]
for l in lines[start + 1:end]:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append(cur); cur = [m.group(1), []]
        continue
    cur[1].append(s.split(";")[0].strip())
blocks.append(cur)
order = {b[0]: k for k, b in enumerate(blocks)}
heads = set()
for k, (name, ins) in enumerate(blocks):
    for x in ins:
        m = re.match(r"s_cbranch\S*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", x)
        if m:
            t = m.group(1) or m.group(2)
            if t in order and order[t] <= k:
                heads.add(t)


def classify(x):
    op = x.split()[0]
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_atomic") or op.startswith("buffer_atomic") or op.startswith("flat_atomic"):
        return "atom"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load") or op.startswith("scratch_load"):
        return "vld"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store") or op.startswith("scratch_store"):
        return "vst"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "sld"
    op = re.sub(r"_e(32|64)$", "", op)
    if op in ("v_exp_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return "other"


tot = {}
for name, ins in blocks:
    c = {}
    for x in ins:
        k = classify(x)
        c[k] = c.get(k, 0) + 1
        tot[k] = tot.get(k, 0) + 1
    if len(ins) >= minn or name == dump:
        print(f"{name:12s}{'*' if name in heads else ' '} n={len(ins):5d} " + " ".join(f"{k}={c[k]}" for k in
              ("valu", "trans", "salu", "mfma", "lds", "vld", "vst", "atom", "wait", "sld") if k in c))
    if name == dump:
        for x in ins:
            print("      ", x)
print("total", " ".join(f"{k}={v}" for k, v in sorted(tot.items())))
