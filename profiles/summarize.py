"""Turn the rocprofv3 databases written by profiles/collect.sh into the committed summaries:
  <tag>_bench_kernel_stats.csv   per-kernel calls / total / average duration (kernel trace)
  <tag>_hbm_traffic.json         per-kernel HBM bytes per launch from FETCH_SIZE (x2 on gfx950 for wide
                                 streaming reads, MI355X_MICROARCH.md) + WRITE_SIZE, separate PMC passes
  <tag>_mfma_counters.json       per-kernel SQ_INSTS_VALU / SQ_INSTS_MFMA / MFMA busy cycles
usage: python profiles/summarize.py gpurun_out/prof_r01 r01 [outdir]   (collect.sh runs it on the GPU box and keeps
only the summaries: the databases are tens of MB)
"""
import collections
import csv
import json
import os
import re
import sqlite3
import sys

src, tag = sys.argv[1], sys.argv[2]
here = sys.argv[3] if len(sys.argv) > 3 else os.path.dirname(os.path.abspath(__file__))   # output directory


def db(sub):
    d = os.path.join(src, sub)
    f = [x for x in os.listdir(d) if x.endswith(".db")][0]
    return sqlite3.connect(os.path.join(d, f))


def have(sub):
    return os.path.isdir(os.path.join(src, sub))


def mangled_of(con):
    """display name -> mangled name.  rocprofv3's own demangler garbles kernels with a __bf16 / _Float16 template
    argument AND further arguments behind it ("conv16_tile_kernel<bool _Accum, int, ELb1EL, bool, E>"): the summaries
    are keyed by the MANGLED name of the symbol table, demangled here."""
    try:
        return {d: m for m, d in con.execute("select kernel_name, display_name from kernel_symbols") if "DF16" in m}
    except sqlite3.Error:
        return {}


def demangle(name):
    """_ZN12_GLOBAL__N_118conv16_tile_kernelIDF16bLi6EEEvNS_10Tile16ArgsE -> conv16_tile_kernel<bf16, 6> (rocprofv3 leaves the
    kernels with a __bf16 / _Float16 template argument mangled)."""
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    base = name[m.end():m.end() + n]
    rest = name[m.end() + n:]
    args = []
    if rest.startswith("I"):
        for mm in re.finditer(r"DF16b|DF16_|Li\d+E|Lb[01]E", rest.split("EEv")[0]):
            t = mm.group(0)
            args.append("bf16" if t == "DF16b" else "f16" if t == "DF16_" else t[2:-1] if t[1] == "i" else ("true" if t[2] == "1" else "false"))
    return base + ("<" + ", ".join(args) + ">" if args else "")


def short(name):
    name = demangle(name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def family(name):
    """conv_mfma_kernel<4, 6, 1, true> -> conv_mfma_kernel<4, 6> (all row-base / staging instantiations)."""
    m = re.match(r"(conv_mfma_kernel)<(\d+), (\d+),", name)
    if m:
        return "%s<%s, %s>" % m.groups()
    m = re.match(r"(wino_conv_kernel)<(\d+),", name)   # <NT, tile geometry> -> <NT>
    if m:
        return "%s<%s>" % m.groups()
    m = re.match(r"(conv16_mfma_kernel)<[^,]+, (\d+)", name)   # <dtype, MT, ...>
    if m:
        return "%s<%s>" % m.groups()
    # conv16_tile_kernel<T, NT, X16, Y16>, conv16_head_kernel<T, CQ, CK, X16, Y16>, wgrad16_kernel<T, MC, LGRW, X16, PL1>:
    # the storage variants are different kernels as far as HBM bytes go — kept apart; the operand type is dropped
    m = re.match(r"(conv16_tile_kernel|conv16_head_kernel|wgrad16_kernel|wgrad16_1x1_kernel)<[^,]+, (.*)>$", name)
    if m:
        return "%s<%s>" % m.groups()
    m = re.match(r"(conv_wgrad_mfma_kernel)<", name)
    return m.group(1) if m else name


def kernel_rows(sub, cols):
    con = db(sub)
    mg = mangled_of(con)
    agg = collections.OrderedDict()
    for r in con.execute("select name, end - start from kernels"):
        n = mg.get(r[0], r[0])
        e = agg.setdefault(n, [0, 0, 1 << 62, 0])
        e[0] += 1
        e[1] += r[1]
        e[2] = min(e[2], r[1])
        e[3] = max(e[3], r[1])
    out = [(n, c, t, t / c, mn, mx) for n, (c, t, mn, mx) in agg.items()]
    out.sort(key=lambda r: -r[2])
    return [r[:cols] for r in out]


rows = kernel_rows("trace", 6) if have("trace") else []
total = sum(r[2] for r in rows)
with open(os.path.join(here, tag + "_bench_kernel_stats.csv"), "w", newline="") if rows else open(os.devnull, "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, c, t, a, mn, mx in rows:
        w.writerow([short(n), c, t, "%.1f" % a, "%.3f" % (100.0 * t / total), mn, mx])
    fam = collections.OrderedDict()
    for n, c, t, a, mn, mx in rows:
        k = family(short(n))
        if k != short(n):
            e = fam.setdefault(k, [0, 0])
            e[0] += c
            e[1] += t
    w.writerow([])
    w.writerow(["# families (all template instantiations of one tile shape)"])
    for k, (c, t) in fam.items():
        w.writerow([k, c, t, "%.1f" % (t / c), "%.3f" % (100.0 * t / total), "", ""])


def kernel_csv(sub, name, note):
    rows = kernel_rows(sub, 4)
    total = sum(r[2] for r in rows)
    fam = collections.OrderedDict()
    for n, c, t, a in rows:
        e = fam.setdefault(family(short(n)), [0, 0])
        e[0] += c
        e[1] += t
    with open(os.path.join(here, name), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["# " + note])
        w.writerow(["Family", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, c, t, "%.1f" % (t / c), "%.3f" % (100.0 * t / total)])


if have("trace16"):
    kernel_csv("trace16", tag + "_bench_bf16_kernel_stats.csv",
               "bench.py --dtype bf16 (U-Net MFMA operands in bf16): kernel trace of 3 warm-up + 10 timed + event steps")
if have("infer"):
    kernel_csv("infer", tag + "_infer4096_kernel_stats.csv",
               "scratch/infer_prof.py 4096: three filled 4096x4096 inferences + NMS (first one includes warm-up)")


def counters(sub):
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    con = db(sub)
    mg = mangled_of(con)
    for k, cn, v in con.execute("select kernel_name, counter_name, value from counters_collection"):
        e = out[family(short(mg.get(k, k)))][cn]
        e[0] += v
        e[1] += 1
    return out


def traffic_of(fsub, wsub):
    fetch, write = counters(fsub), counters(wsub)
    traffic = {}
    for k in sorted(set(fetch) | set(write)):
        if not any(t in k for t in ("conv", "wino", "wgrad", "act_bwd", "bn_", "nms", "unrot", "rot4", "maxpool", "head", "adam")):
            continue      # library kernels only (the stock ATen elementwise kernels are not this repo's)
        fk = fetch.get(k, {}).get("FETCH_SIZE", [0.0, 1])
        wk = write.get(k, {}).get("WRITE_SIZE", [0.0, 1])
        fkb, wkb = fk[0] / max(fk[1], 1), wk[0] / max(wk[1], 1)
        traffic[k] = {"launches": int(max(fk[1], wk[1])), "FETCH_SIZE_KB_avg": fkb, "WRITE_SIZE_KB_avg": wkb,
                      "hbm_bytes_per_launch": (2.0 * fkb + wkb) * 1024.0}
    return traffic


def previous(name):
    try:
        with open(os.path.join(here, tag + name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


prev_t, prev_m = previous("_hbm_traffic.json"), previous("_mfma_counters.json")
traffic = traffic_of("fetch", "write") if have("fetch") and have("write") else prev_t.get("kernels", {})
traffic16 = traffic_of("fetch16", "write16") if have("fetch16") and have("write16") else prev_t.get("kernels_bf16_step", {})
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402  (bench.py trusts this file only for the same kernel sources)

json.dump({"kernel_source_hash": kernel_source_hash(),
           "note": "per-launch averages over the profiled bench run; FETCH_SIZE doubled (gfx950 counts 128-B "
                   "requests at 64 B for wide streaming reads), WRITE_SIZE as is; separate PMC passes",
           "kernels": traffic,
           "kernels_bf16_step": traffic16,
           "bf16_note": "kernels_bf16_step: the same passes on bench.py --dtype bf16 (16-bit operands AND 16-bit activation "
                        "tensors between the U-Nets' layers)"},
          open(os.path.join(here, tag + "_hbm_traffic.json"), "w"), indent=1)

if have("mfma"):
    mf = counters("mfma")
    mfj = {k: {cn: {"sum": v[0], "launches": v[1]} for cn, v in d.items()} for k, d in mf.items() if "conv" in k or "wino" in k}
else:
    mfj = {k: v for k, v in prev_m.items() if k != "bf16_step"}
if not have("mfma16") and "bf16_step" in prev_m:
    mfj["bf16_step"] = prev_m["bf16_step"]
if have("mfma16"):
    mfj["bf16_step"] = {k: {cn: {"sum": v[0], "launches": v[1]} for cn, v in d.items()} for k, d in counters("mfma16").items()
                        if "conv" in k or "wgrad" in k or "wino" in k}
json.dump(mfj, open(os.path.join(here, tag + "_mfma_counters.json"), "w"), indent=1)
for name in ("bench.json", "bench_under_rocprof.json"):
    if os.path.exists(os.path.join(src, name)) and os.path.getsize(os.path.join(src, name)):
        open(os.path.join(here, "%s_%s" % (tag, name)), "w").write(open(os.path.join(src, name)).read())
print("wrote summaries for", tag)
