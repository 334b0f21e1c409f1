"""Launches, busy time and host round trips per time step from a rocprofv3 kernel trace of
`bench.py --mode timestep --kBT 1 ...` (steps are delimited by the noise kernel k_normal).
    python tools/step_launches.py <kernel_trace.csv> [title]   -> markdown on stdout
A gap of >= 15 us between two kernels is counted as a host round trip (a stream drain, a Python-level call boundary);
shorter gaps are launch spacing.  Times are those of a PROFILED run: gaps carry the profiler's per-dispatch overhead."""
import csv, re, sys, collections


def name(r):
    s = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    return re.split(r"[(<]", s)[0]


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cut = [i for i, r in enumerate(rows) if name(r) == "k_normal"]
print("# %s\n" % (sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]))
print("| step | launches | span, ms | kernels busy, ms | of it products (pair kernels + slab sums), ms / count | round trips (gaps >= 15 us) | their sum, ms | launch spacing (gaps < 15 us), ms |")
print("|---|---|---|---|---|---|---|---|")
acc = collections.Counter(); dur = collections.Counter()
for k, (a, b) in enumerate(zip(cut[:-1], cut[1:])):
    sub = rows[a:b]
    span = (int(rows[b]["Start_Timestamp"]) - int(sub[0]["Start_Timestamp"])) / 1e6
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in sub]
    gaps = [(int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e6 for x, y in zip(sub[:-1], sub[1:])]
    prod = [x for r, x in zip(sub, d) if name(r).startswith("k_apply_M") or name(r) == "k_reduce_sym"]
    npr = sum(1 for r in sub if name(r).startswith("k_apply_M"))
    big = [g for g in gaps if g >= 0.015]
    print("| %d | %d | %.2f | %.2f | %.2f / %d | %d | %.2f | %.2f |" % (k, len(sub), span, sum(d), sum(prod), npr, len(big), sum(big),
                                                                 sum(g for g in gaps if g < 0.015)))
    if k == len(cut) - 2:
        for r, x in zip(sub, d):
            acc[name(r)] += 1; dur[name(r)] += x
print("\nkernels of the last step:\n\n| kernel | launches | total, us |\n|---|---|---|")
for n_, c in sorted(acc.items(), key=lambda kv: -dur[kv[0]]):
    print("| %s | %d | %.0f |" % (n_, c, dur[n_] * 1e3))
