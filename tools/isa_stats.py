#!/usr/bin/env python3
"""Per-pair instruction counts of the matvec kernels, taken from the gfx950 assembly of the SAME sources
librbl.so is built from (rigid_body_light_amd/build.py runs this after every kernel build and leaves the
result next to the library: librbl.isa.json).  bench.py prices its roofline with these EXECUTED counts
(frac <= 1 by construction) instead of the reference-arithmetic 204 flop per ordered pair.

For every k_apply_M_sym / k_apply_M_sym2 / k_apply_M_sym4 instantiation the hot block is the far-tile
systolic sweep: the basic block with the most v_rsq_f64 that has ds_add_f64 column sums and no division.
Pairs per trip of that block = v_rsq_f64 / (2 with the wall term, 1 without).

usage: isa_stats.py kernels.s out.json
"""
import json
import re
import sys


def blocks_of(lines, start, end):
    out = []
    cur = ["entry", {}]
    out.append(cur)
    for l in lines[start + 1:end]:
        if re.match(r"^\.LBB\d+_\d+:", l):
            cur = [l.split(":")[0], {}]
            out.append(cur)
            continue
        t = l.strip().split(" ")[0].split("\t")[0] if l.strip() else ""
        if t and not t.startswith(";") and not t.startswith("."):
            cur[1][t] = cur[1].get(t, 0) + 1
    return out


def classify(ops):
    g = lambda pred: sum(v for k, v in ops.items() if pred(k))
    fma = g(lambda k: k.startswith(("v_fma_f64", "v_fmac_f64")))
    mul = g(lambda k: k.startswith("v_mul_f64"))
    add = g(lambda k: k.startswith("v_add_f64"))
    trans = g(lambda k: k.startswith(("v_rsq_f64", "v_rcp_f64", "v_sqrt_f64")))
    div = g(lambda k: k.startswith("v_div_"))
    mfma = g(lambda k: k.startswith("v_mfma"))
    f64 = g(lambda k: k.startswith("v_") and "f64" in k and not k.startswith("v_mfma"))
    valu = g(lambda k: k.startswith("v_") and not k.startswith("v_mfma"))
    lds = g(lambda k: k.startswith("ds_"))
    salu = g(lambda k: k.startswith("s_") and not k.startswith(("s_waitcnt", "s_nop", "s_barrier")))
    return {"fma": fma, "mul": mul, "add": add, "trans": trans, "div": div, "mfma": mfma, "f64": f64,
            "valu": valu, "valu_other": valu - f64, "lds": lds, "salu": salu,
            "rsq": g(lambda k: k.startswith("v_rsq_f64")), "ds_add": g(lambda k: k.startswith("ds_add_f64"))}


def isa_hash(lines, start, end):
    """sha256 of the instruction text of one function: comments and directives dropped, basic-block labels renumbered
    without the function's index in the file -- unrelated edits elsewhere in the source leave it unchanged"""
    import hashlib
    h = hashlib.sha256()
    for l in lines[start + 1:end]:
        t = l.split(";")[0].strip()
        if not t or (t.startswith(".") and not t.startswith(".LBB")):
            continue
        h.update(re.sub(r"\.LBB\d+_(\d+)", r".LBB_\1", " ".join(t.split())).encode())
        h.update(b"\n")
    return h.hexdigest()


def main():
    src, dst = sys.argv[1], sys.argv[2]
    lines = open(src).read().split("\n")
    res = {}
    instances = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN\S*?(k_apply_M_sym\d?)ILb([01])ELi(\d)E((?:Li\d+E)*)E\S*):", l)
        if not m:
            continue
        kern, wall, ni = m.group(2), m.group(3) == "1", int(m.group(4))
        extra = [int(x) for x in re.findall(r"Li(\d+)E", m.group(5))]      # SW [, PREC]
        relaxed = len(extra) >= 2 and extra[1] == 1
        end = next(k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end"))
        instances["%s<%s,%d%s>" % (kern, "true" if wall else "false", ni, "".join(",%d" % x for x in extra))] = isa_hash(lines, i, end)
        best = None
        if relaxed:      # the packed single-precision sweep: v_rsq_f32, two pairs per packed instruction, column sums by ds_add_f64
            for name, ops in blocks_of(lines, i, end):
                g = lambda pred: sum(v for k, v in ops.items() if pred(k))
                rsq32 = g(lambda k: k.startswith("v_rsq_f32"))
                if rsq32 == 0 or g(lambda k: k.startswith("ds_add_f")) == 0 or g(lambda k: k.startswith("v_rsq_f64")):
                    continue
                if best is None or rsq32 > best[1]:
                    best = (name, rsq32, ops)
            if best is None:
                continue
            ops = best[2]
            g = lambda pred: sum(v for k, v in ops.items() if pred(k))
            pairs = best[1] / (2.0 if wall else 1.0)
            pk_fma, pk_mul, pk_add = g(lambda k: k.startswith("v_pk_fma_f32")), g(lambda k: k.startswith("v_pk_mul_f32")), g(lambda k: k.startswith("v_pk_add_f32"))
            f32 = g(lambda k: k.startswith(("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32")))
            valu = g(lambda k: k.startswith("v_"))
            res["%s<%s,%d,relaxed>" % (kern, "true" if wall else "false", ni)] = {
                "block": best[0], "unordered_pairs_per_trip": pairs,
                "per_unordered_pair": {"valu": valu / pairs, "pk_fma": pk_fma / pairs, "pk_mul": pk_mul / pairs, "pk_add": pk_add / pairs,
                                       "rsq_f32": best[1] / pairs, "f32_scalar": f32 / pairs, "lds": g(lambda k: k.startswith("ds_")) / pairs,
                                       "flop": (4 * pk_fma + 2 * pk_mul + 2 * pk_add + f32 + best[1]) / pairs}}
            continue
        for name, ops in blocks_of(lines, i, end):
            c = classify(ops)
            if c["rsq"] == 0 or c["ds_add"] == 0 or c["div"] or c["trans"] != c["rsq"]:
                continue
            if best is None or c["rsq"] > best[1]["rsq"]:
                best = (name, c)
        if best is None:
            continue
        c = best[1]
        pairs = c["rsq"] / (2.0 if wall else 1.0)
        per = {k: c[k] / pairs for k in ("fma", "mul", "add", "trans", "f64", "valu", "valu_other", "lds", "salu")}
        per["flop"] = 2 * per["fma"] + per["mul"] + per["add"] + per["trans"]   # a transcendental counted as ONE flop
        res["%s<%s,%d>" % (kern, "true" if wall else "false", ni)] = {
            "block": best[0], "unordered_pairs_per_trip": pairs, "per_unordered_pair": per}
    # the wave-unit kernel of mid-size systems, k_apply_M_symw<WALL, IW>: its sweep is a loop of a few basic blocks (the rare overlap
    # branch splits it), column sums rotating through the lanes by v_mov_b32_dpp wave_rol:1 -- summed from the loop header to the
    # block that branches back to it; pair steps per trip = v_rsq_f64 / (2 with the wall term, 1 without)
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN\S*?14k_apply_M_symwILb([01])ELi(\d+)ELi(\d+)EE\S*):", l)
        if not m:
            continue
        wall = m.group(2) == "1"
        ni = int(m.group(3))
        end = next(k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end"))
        name_k = ("k_apply_M_symw<%s>" if ni == 1 else "k_apply_M_symw<%%s,%d>" % ni) % ("true" if wall else "false")
        instances[name_k] = isa_hash(lines, i, end)
        labels, raw = [], []
        for k in range(i + 1, end):
            if re.match(r"^\.LBB\d+_\d+:", lines[k]):
                labels.append(lines[k].split(":")[0]); raw.append([])
            elif raw:
                raw[-1].append(lines[k])
        for b, body in enumerate(raw):
            txt = "\n".join(body)
            back = re.search(r"s_cbranch_scc[01] (\.LBB\d+_\d+)", txt)
            if "wave_rol" not in txt or not back or back.group(1) not in labels[:b + 1]:
                continue
            h = labels.index(back.group(1))
            ops = {}
            for body2 in raw[h:b + 1]:
                for ln in body2:
                    t = ln.strip().split(" ")[0].split("\t")[0] if ln.strip() else ""
                    if t and not t.startswith(";") and not t.startswith("."):
                        ops[t] = ops.get(t, 0) + 1
            c = classify(ops)
            pairs = c["rsq"] / (2.0 if wall else 1.0)
            if pairs <= 0:
                continue
            per = {k: c[k] / pairs for k in ("fma", "mul", "add", "trans", "f64", "valu", "valu_other", "lds", "salu")}
            per["dpp_mov"] = sum(v for k, v in ops.items() if k.startswith("v_mov_b32_dpp")) / pairs
            per["flop"] = 2 * per["fma"] + per["mul"] + per["add"] + per["trans"]
            if name_k not in res or res[name_k]["unordered_pairs_per_trip"] < pairs:   # (NI = 2: the one-row sweep of the lane's own second tile comes first)
                res[name_k] = {"block": "%s .. %s" % (labels[h], labels[b]), "unordered_pairs_per_trip": pairs, "per_unordered_pair": per}
    # the ordered-rows kernel k_apply_M<WALL> (row-sharded multi-GPU split): its sweep is a run of one-pair head blocks (distance,
    # rsq, three ds_read_b128 broadcasts of the staged j blob) followed by ONE body block that finishes those pairs together
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN\S*?9k_apply_MILb([01])EE\S*):", l)
        if not m:
            continue
        wall = m.group(2) == "1"
        end = next(k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end"))
        instances["k_apply_M<%s>" % ("true" if wall else "false")] = isa_hash(lines, i, end)
        blks = [(name, classify(ops)) for name, ops in blocks_of(lines, i, end)]
        for b, (name, c) in enumerate(blks):
            if not (c["rsq"] == 1 and c["lds"] == 3 and not c["div"]):
                continue
            nhead = 0
            while b + nhead < len(blks) and blks[b + nhead][1]["rsq"] == 1 and blks[b + nhead][1]["lds"] == 3:
                nhead += 1
            if b + nhead >= len(blks):
                break
            body = blks[b + nhead][1]
            if body["lds"] or body["div"] or body["f64"] < 8 * nhead:
                break
            per = {k: c[k] + body[k] / float(nhead) for k in ("fma", "mul", "add", "trans", "f64", "valu", "valu_other", "lds", "salu")}
            per["flop"] = 2 * per["fma"] + per["mul"] + per["add"] + per["trans"]
            res["k_apply_M<%s>" % ("true" if wall else "false")] = {"block": "%s x%d + %s" % (name, nhead, blks[b + nhead][0]),
                                                                     "ordered_pairs_per_trip": nhead, "per_ordered_pair": per}
            break
    import hashlib
    import os
    h = hashlib.sha256()
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rigid_body_light_amd", "csrc")
    for f in ("rbl_kernels.hip", "rbl_pair.hpp"):
        h.update(open(os.path.join(csrc, f), "rb").read())
    json.dump({"source": "hipcc -S --offload-device-only of csrc/rbl_kernels.hip (same flags as librbl.so)",
               "kernel_source_sha256": h.hexdigest(), "kernels": res, "instance_isa_sha256": instances},
              open(dst, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items()):
        if "per_ordered_pair" in v:
            p = v["per_ordered_pair"]
            print("%-28s %s: %.1f VALU (%.1f f64: %.1f fma %.1f mul %.1f add %.1f trans) %.1f LDS -> %.0f flop / ORDERED pair"
                  % (k, v["block"], p["valu"], p["f64"], p["fma"], p["mul"], p["add"], p["trans"], p["lds"], p["flop"]))
            continue
        p = v["per_unordered_pair"]
        if "pk_fma" in p:
            print("%-28s %s: %.1f VALU (%.1f pk_fma %.1f pk_mul %.1f pk_add %.1f rsq_f32) %.1f LDS / unordered pair"
                  % (k, v["block"], p["valu"], p["pk_fma"], p["pk_mul"], p["pk_add"], p["rsq_f32"], p["lds"]))
            continue
        print("%-28s %s: %.1f VALU (%.1f f64: %.1f fma %.1f mul %.1f add %.1f trans) %.1f LDS -> %.0f flop / unordered pair"
              % (k, v["block"], p["valu"], p["f64"], p["fma"], p["mul"], p["add"], p["trans"], p["lds"], p["flop"]))


if __name__ == "__main__":
    main()
