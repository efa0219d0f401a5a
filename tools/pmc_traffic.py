"""HBM-side bytes per launch of bench.py's roofline kernels from two rocprofv3 --pmc passes over tools/gemm_pmc.py
(host-side): FETCH_SIZE (KB; x2 on gfx950 for wide 16-B/lane reads, MI355X_MICROARCH.md) + WRITE_SIZE (KB).
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> > profiles/r02_pmc_traffic.json"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

# product mode 1 (split bf16, csrc/gemm_x3.hip / gemm_x3_bfrag.hip): forward = gemm_x3_bfrag_kernel<true, ...> (or gemm_x3_kernel<false, true, ...> with RN_X3_BFRAG=0), backward = data gradient <false, false, ...> +
# weight gradient <true, true, ...> (two launches); product mode 0: the fp32 matrix-core kernels
KERNELS_X3 = {"fwd_products": ("gemm_x3_kernel<false, true", "gemm_x3_bfrag_kernel<true"), "bwd_products": ("gemm_x3_kernel<false, false", "gemm_x3_kernel<true, true"),
              "group_norm": ("stem_conv_fwd_kernel", "gn_apply_rows_kernel")}
KERNELS_F32 = {"fwd_products": ("conv_fwd_kernel",), "bwd_products": ("conv_bwd_kernel",),
               "group_norm": ("stem_conv_fwd_kernel", "gn_apply_rows_kernel")}


def per_kernel(directory, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main(fetch_dir, write_dir):
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    out, detail = {}, {}
    x3 = any("gemm_x3_kernel" in k for k in fetch)
    for key, names in (KERNELS_X3 if x3 else KERNELS_F32).items():
        total = 0.0
        for n in names:
            # the demangled name STARTS with the kernel's name ("conv_fwd_kernel" is also a substring of "stem_conv_fwd_kernel")
            mine = lambda k: re.match(r"(void\s+)?(\(anonymous namespace\)::)?%s(\b|,)" % re.escape(n), k) is not None
            f = [v for k, vals in fetch.items() if mine(k) for v in vals]
            w = [v for k, vals in write.items() if mine(k) for v in vals]
            if not f or not w:
                continue
            fk, wk = sum(f) / len(f), sum(w) / len(w)
            detail["%s/%s" % (key, n)] = {"FETCH_SIZE_KB_avg": round(fk, 1), "WRITE_SIZE_KB_avg": round(wk, 1), "dispatches": len(f)}
            total += (2.0 * fk + wk) * 1024.0       # gfx950: FETCH_SIZE reports half of the bytes of wide streaming reads
        out[key] = int(total) if total else None
    out["_detail"] = detail
    # what tools/gemm_pmc.py launched: bench.py compares this with its own workload and drops `traffic` when they differ
    out["_shape"] = {"points": 36, "tiles": 2 * sum(((s + 3) // 4) ** 2 for s in (64, 32, 16, 8, 4)), "cin": 256, "cout": 256, "batch": 2, "image": 512}
    out["_product_mode"] = 1 if x3 else 0
    # ... and the kernel sources it measured: a kernel edit that keeps the shape must not leave this figure in the driver's line
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from src_hash import kernel_source_hash, FILES
    out["_kernel_source_sha"] = kernel_source_hash()
    out["_kernel_source_files"] = list(FILES)
    out["_note"] = "bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes), averaged over the dispatches of tools/gemm_pmc.py"
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
