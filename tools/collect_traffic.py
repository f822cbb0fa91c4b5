"""HBM traffic per launch of the ASDNet kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), collected and
corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes (separate --pmc passes; FETCH_SIZE x2 on gfx950).

On the GPU box (each pass is its own rocprofv3 run, no tracing options):
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 tools/time_asdnet.py 2000 3
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 tools/time_asdnet.py 2000 3
  python3 tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 2000 > gpurun_out/traffic_conv2.json
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def per_kernel(d, counter):
    out = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    # a kernel launched fewer than three times is not part of the measured forwards (the calibration pass at asd_load_weights)
    return {k: sum(v) / len(v) for k, v in out.items() if len(v) >= 3}


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", name).strip()


def main():
    fetch_dir, write_dir, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    allk = {}
    for k in fetch:
        if "k_conv_mfma" in k or "k_conv_x3" in k or "k_fc_mfma" in k or "k_fc_x" in k or "k_l2norm" in k:
            f_kb, w_kb = fetch[k], write.get(k, 0.0)
            allk[short(k)] = {"FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb, "hbm_bytes_per_launch": int(2.0 * f_kb * 1024 + w_kb * 1024)}
    fused = [k for k in allk if k.startswith("k_conv_x3<32, 32, 32")] or [k for k in allk if k.startswith("k_conv_mfma<32, 32, 32")]
    if not fused:
        raise SystemExit("fused conv2 kernel not found in the counter files: " + ", ".join(allk))
    k2 = allk[fused[0]]
    out = {"kernel": fused[0] + " (ASDNet input_norm+conv1+conv2 fused), N=%d patches" % n, "n_patches": n,
           "FETCH_SIZE_KB": k2["FETCH_SIZE_KB"], "WRITE_SIZE_KB": k2["WRITE_SIZE_KB"], "fetch_correction": 2.0,
           "hbm_bytes_per_launch": k2["hbm_bytes_per_launch"],
           # u8 patch in, conv2 activation out at 4 B per element (f32 NHWC, or the fp16 piece pair of the pair format), weights once
           # (f32 image 36,992 B x 4; the split image holds 4-6 B per weight)
           "algorithmic_bytes_per_launch": n * (1024 + 32 * 32 * 32 * 4) + (9 * 32 * 32 * 6 + 320 * 4 if fused[0].startswith("k_conv_x3") else 36992 * 4),
           "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), tools/time_asdnet.py %d 3; FETCH_SIZE doubled per "
                     "MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)" % n,
           "all_kernels": allk}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
