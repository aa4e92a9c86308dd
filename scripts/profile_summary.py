"""Turn rocprofv3 output directories into the small summaries committed under profiles/.

    python scripts/profile_summary.py stats  <dir with *_kernel_stats.csv>  <out.csv>
    python scripts/profile_summary.py pmc    <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass> <out.json> "<command>"

stats: copies the kernel_stats table of the largest trace (the bench process), shortening template names.
pmc:   per-kernel HBM bytes per launch.  FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; on gfx950 FETCH_SIZE
       reports half the bytes of wide coalesced / LDS-DMA reads (MI355X_MICROARCH.md, HBM section), so it is
       doubled; WRITE_SIZE is taken as is.  The two counters come from separate passes (they do not fit one).
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    name = name.replace("unsigned short", "bf16")
    return name


FAMILIES = (("conv_igemm_kernel", "conv_igemm"), ("conv_pws_kernel", "conv_igemm"), ("conv_chain_kernel", "conv_igemm"), ("conv_bwd3_kernel", "conv_igemm"), ("conv_pp_kernel", "conv_igemm"), ("conv_c64_kernel", "conv_igemm"), ("conv_dfold_kernel", "conv_igemm"), ("gram_kernel", "gram"), ("wgrad_pp_kernel", "conv_wgrad"), ("wgrad3x3_patch_kernel", "conv_wgrad"), ("wgrad3x3_wide_kernel", "conv_wgrad"), ("wgrad_ring_kernel", "conv_wgrad"), ("wgrad_kernel", "conv_wgrad"),
                 ("bn_act_fwd2_kernel", "bn_act_fwd"), ("bn_act_fwd_kernel", "bn_act_fwd"), ("bn_bwd_apply2_kernel", "bn_bwd_apply"),
                 ("bn_bwd_apply_kernel", "bn_bwd_apply"), ("bn_bwd_reduce_kernel", "bn_bwd_reduce"),
                 ("reduce_partials_kernel", "reduce_partials"))


def biggest(pattern):
    files = glob.glob(pattern, recursive=True)
    if not files:
        raise SystemExit("no file matches %s" % pattern)
    return max(files, key=os.path.getsize)


def family(name):
    """conv_igemm_kernel<bf16, 256, 128, 3, 6, false, true> -> conv_igemm (the names bench.py's table uses: every
    convolution / data-gradient launch, ring or streaming kernel, is "conv_igemm"; every weight gradient "conv_wgrad")."""
    base = name.split("<")[0]
    for a, b in FAMILIES:
        if base == a:
            return b
    return base


def stats(src, out):
    f = biggest(os.path.join(src, "**", "*_kernel_stats.csv"))
    fam = {}
    with open(f) as fh, open(out, "w", newline="") as oh:
        rd = csv.DictReader(fh)
        wr = csv.writer(oh)
        wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rd:
            wr.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], "%.1f" % float(r["AverageNs"]), r["Percentage"],
                         r["MinNs"], r["MaxNs"]])
            e = fam.setdefault(family(short(r["Name"])), [0, 0.0])
            e[0] += int(r["Calls"])
            e[1] += float(r["TotalDurationNs"])
        # the families bench.py's live table uses (every forward / data-gradient convolution launch is "conv_igemm",
        # every weight gradient "conv_wgrad"): calls, total and average duration over all kernels of the family
        wr.writerow([])
        wr.writerow(["Family", "Calls", "TotalDurationNs", "AverageNs"])
        for k, (n, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            wr.writerow(["family:" + k, n, "%.0f" % t, "%.1f" % (t / max(n, 1))])
    print("wrote", out, "from", f)


def per_kernel(src, counter):
    f = biggest(os.path.join(src, "**", "*_counter_collection.csv"))
    acc = {}
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def pmc(fetch_dir, write_dir, out, command):
    rd, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    fam = {}
    for k in sorted(set(rd) | set(wr)):
        n = max(rd.get(k, [0, 0])[0], wr.get(k, [0, 0])[0])
        e = fam.setdefault(family(k), {"launches": 0, "read": 0.0, "write": 0.0, "variants": {}})
        rb = 2.0 * 1024.0 * rd.get(k, [0, 0.0])[1]
        wb = 1024.0 * wr.get(k, [0, 0.0])[1]
        e["launches"] += n
        e["read"] += rb
        e["write"] += wb
        e["variants"][k] = {"launches": n, "hbm_bytes_per_launch": (rb + wb) / max(n, 1)}
    kernels = {}
    for k, e in fam.items():
        n = max(e["launches"], 1)
        kernels[k] = {"launches": e["launches"], "read_bytes_per_launch": e["read"] / n, "write_bytes_per_launch": e["write"] / n,
                      "hbm_bytes_per_launch": (e["read"] + e["write"]) / n}
        if len(e["variants"]) > 1:
            kernels[k]["variants"] = e["variants"]
    import hashlib
    import os
    h = hashlib.sha256()   # same digest as bench.py csrc_digest(): ties this summary to the kernels it was measured on
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multimodal-active-ai_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            with open(os.path.join(d, fn), "rb") as fh:
                h.update(fn.encode())
                h.update(fh.read())
    total = sum(e["read"] + e["write"] for e in fam.values())
    doc = {"command": command, "csrc_digest": h.hexdigest()[:16], "total_hbm_bytes_all_dispatches": total,
           "correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced / LDS-DMA reads, MI355X_MICROARCH.md "
                         "HBM section); WRITE_SIZE as is; both in KiB per dispatch; separate passes per counter",
           "kernels": kernels}
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print("wrote", out)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "")
    else:
        raise SystemExit(__doc__)
