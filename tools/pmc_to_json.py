#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes of `bench.py` into the small JSON file bench.py reads for its `roofline.traffic` field:
HBM bytes per launch of the dominant kernel = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> B; on gfx950 FETCH_SIZE reports half
of the bytes of wide coalesced reads -- MI355X_MICROARCH.md, section HBM), plus the SQ counters for the issue-slot view.
The file records the hash of the kernel sources the profiled library was built from; bench.py ignores it when the
sources have changed since.

usage: pmc_to_json.py <kernel substring, e.g. "k_apply_M_sym<true, 2"> <config> out.json dir1 [dir2 ...]"""
import collections, csv, glob, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
key, config, out = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(list)
dur = []
for d in sys.argv[4:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
avg = {k: sum(v) / len(v) for k, v in agg.items()}
h = hashlib.sha256()
for f in ("rbl_kernels.hip", "rbl_pair.hpp"):
    h.update(open(os.path.join(ROOT, "rigid_body_light_amd", "csrc", f), "rb").read())
inst = key.replace(" ", "")
inst = inst if inst.endswith(">") else inst + ">"
try:   # instruction-text hash of the profiled instance (tools/isa_stats.py via the build): what bench.py compares
    isa_sha = json.load(open(os.path.join(ROOT, "rigid_body_light_amd", "librbl.isa.json")))["instance_isa_sha256"].get(inst)
except (OSError, KeyError, ValueError):
    isa_sha = None
doc = {"kernel": None, "kernel_match": key, "config": config, "kernel_source_sha256": h.hexdigest(),
       "kernel_instance": inst, "kernel_isa_sha256": isa_sha,
       "counters_avg_per_dispatch": avg, "dispatches": {k: len(v) for k, v in agg.items()},
       "source": "rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py (separate passes), averaged per dispatch by tools/pmc_to_json.py"}
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    doc["hbm_bytes_per_launch"] = (2.0 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024.0
    doc["hbm_read_bytes"] = 2.0 * avg["FETCH_SIZE"] * 1024.0
    doc["hbm_write_bytes"] = avg["WRITE_SIZE"] * 1024.0
if dur:
    doc["kernel_ms_under_profiler"] = sum(dur) / len(dur)
if "GRBM_GUI_ACTIVE" in avg and dur:
    doc["sustained_clock_ghz"] = avg["GRBM_GUI_ACTIVE"] / 8.0 / (sum(dur) / len(dur) * 1e-3) / 1e9   # counter sums the 8 XCDs
if "SQ_INSTS_VALU" in avg and "GRBM_GUI_ACTIVE" in avg:
    doc["valu_issue_frac_of_sustained_cycles"] = avg["SQ_INSTS_VALU"] * 4.0 / (avg["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
import re
m = re.match(r"(k_[A-Za-z0-9_]+)<(true|false), (\d)", key)
if m:
    doc["kernel"] = "%s<%s,%s>" % (m.group(1), m.group(2), m.group(3))
json.dump(doc, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(doc, indent=1, sort_keys=True))
