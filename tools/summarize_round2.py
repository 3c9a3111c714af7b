# Turn the raw output of tools/profile_round2.sh (gpurun_out/<tag>_*) into the tracked summaries
# under profiles/.  usage: python tools/summarize_round2.py <tag> <out-prefix>
import csv, glob, json, os, re, shutil, sys, collections

tag, out = sys.argv[1], sys.argv[2]
G, P = "gpurun_out", "profiles"


def kname(n):
    m = re.search(r"(k_\w+(<[^>]*>)?)", n)
    return m.group(1) if m else n[:40]


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            per[(row["Dispatch_Id"], kname(row["Kernel_Name"]), row["Counter_Name"])] += float(row["Counter_Value"])
        for (_, k, c), v in per.items():
            acc[k][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items() if k.startswith("k_")}


for cfg in ("C2", "C4", "C3", "C2_mf"):
    f = os.path.join(G, "%s_bench_%s.log" % (tag, cfg))
    if os.path.exists(f):
        lines = [l for l in open(f).read().split("\n") if l.startswith("{")]
        json.dump([json.loads(l) for l in lines], open(os.path.join(P, "%s_bench_%s.json" % (out, cfg)), "w"), indent=1)
for cfg in ("C2", "C4", "C3"):
    for f in glob.glob(os.path.join(G, "%s_prof_%s" % (tag, cfg), "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(P, "%s_kernel_stats_%s.csv" % (out, cfg)))

hbm = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, v in counters(os.path.join(G, "%s_pmc_%s" % (tag, c))).items():
        hbm.setdefault(k, {})[c + "_KB"] = v[c]
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 "
           "--no-cpu-baseline --no-pipeline (two separate passes, tools/profile_round2.sh)",
           "unit": "KB per launch (counter value), mean over the launches of the run", "kernels": hbm,
           "note": "raw FETCH_SIZE (8 B/lane workspace reads; see r01_e_pmc_hbm.json for why no x2 correction is applied)"},
          open(os.path.join(P, out + "_pmc_hbm.json"), "w"), indent=1)
for k, v in hbm.items():
    if k.startswith("k_solve"):
        total = (v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024.0
        json.dump({"k_solve_hbm_bytes_per_launch": total,
                   "source": "profiles/%s_pmc_hbm.json (FETCH_SIZE raw + WRITE_SIZE), fused assemble+solve+select launch %s" % (out, k),
                   "upper_bound_with_fetch_doubled": (2 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024.0},
                  open(os.path.join(P, "traffic.json"), "w"), indent=1)
for solver in ("tw", "mf"):
    merged = {}
    for i in (1, 2, 3):
        for k, v in counters(os.path.join(G, "%s_sq_%s_%d" % (tag, solver, i))).items():
            merged.setdefault(k, {}).update(v)
    for k, v in merged.items():
        if "SQ_WAVE_CYCLES" in v and "SQ_WAIT_ANY" in v:
            v["derived_wait_any_frac_of_wave_cycles"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_BUSY_CYCLES" in v:
            v["derived_mfma_busy_per_busy_cycle"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["SQ_BUSY_CYCLES"]
    json.dump({"command": "SLOD_SOLVE=%s SLOD_FUSE_SELECT=0 SLOD_FUSE_ASSEMBLE=0 rocprofv3 --pmc <8 SQ counters> --kernel-trace -- python3 bench.py "
                          "--steps 2 --warmup 1 --no-cpu-baseline --no-pipeline (three separate passes, stages unfused so that the "
                          "counters are per stage)" % solver,
               "unit": "counter value per launch, summed over all XCDs/SEs; C2, 1024 patches", "kernels": merged},
              open(os.path.join(P, "%s_sq_%s.json" % (out, solver)), "w"), indent=1)
print(open(os.path.join(P, "traffic.json")).read())
