"""Matrix-core utilisation of the ASDNet kernels from two rocprofv3 PMC passes (each its own run, no tracing options):
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_clk -o c -- python3 tools/time_asdnet.py 2000 3
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_mfma -o m -- python3 tools/time_asdnet.py 2000 3
  python3 tools/collect_mfma_util.py gpurun_out/pmc_clk gpurun_out/pmc_mfma > profiles/r01_asdnet_mfma_util.json
SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over the chip (= 64 x the number of v_mfma_f32_32x32x2_f32, or
32 x the number of v_mfma_f32_32x32x16_bf16 of the split-operand kernels: six per 16-deep k-chunk; checked below);
GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md), so busy / (GUI_ACTIVE / 8 x 1024 SIMDs) is the fraction of
SIMD-cycles with the matrix pipe busy, and GUI_ACTIVE / 8 / duration the effective clock."""
import collections
import csv
import glob
import json
import re
import sys

FLOP = {"k_conv_mfma<32, 32, 32": 2 * 9437184, "k_conv_mfma_p<32, 64": 2 * 4718592, "k_conv_mfma<64, 64": 2 * 9437184,
        "k_conv_mfma<64, 128": 2 * 4718592, "k_conv_mfma<128, 128": 2 * 9437184, "k_fc_mfma": 2 * 1048576,
        "k_conv_x3<32, 32, 32": 2 * 9437184, "k_conv_x3<32, 64": 2 * 4718592, "k_conv_x3<64, 64": 2 * 9437184,
        "k_conv_x3<64, 128": 2 * 4718592, "k_conv_x3<128, 128": 2 * 9437184, "k_fc_x3": 2 * 1048576, "k_fc_x2": 2 * 1048576}


def load(d, name):
    out = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                k = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).strip()
                out[k].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main():
    clk, mf = load(sys.argv[1], "GRBM_GUI_ACTIVE"), load(sys.argv[2], "SQ_VALU_MFMA_BUSY_CYCLES")
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    out = {}
    for k, v in clk.items():
        if "mfma" not in k and "k_conv_x3" not in k and "k_fc_x" not in k:
            continue
        if len(v) < 4:
            continue                      # a kernel the measured forwards do not use (the calibration pass at load)
        v, m = v[2:], mf.get(k, [])[2:]   # drop the warm-up launches
        ga = sum(a for a, _ in v) / len(v)
        dur = sum(d for _, d in v) / len(v)
        busy = sum(a for a, _ in m) / max(len(m), 1)
        flop = [f for p, f in FLOP.items() if k.startswith(p)]
        split = "k_conv_x3" in k or "k_fc_x" in k
        # f32 MFMA 32x32x2: 4096 FLOP, 64 cycles; 16-bit MFMA 32x32x16 (or two 16x16x32): 32768 FLOP, 32 cycles; per f32-equivalent
        # chunk the split kernels issue six products (three bf16 pieces) or three (two fp16 pieces: last template argument 2)
        # k_conv_x3<..., FUSE1, NP[, PAIR]>: NP = 2 -> three products; k_fc_x2 likewise
        products = 3 if ("k_fc_x2" in k or re.search(r",\s*2(,\s*(true|false))?>$", k.strip())) else 6
        n_mfma = (products * n * flop[0] / 32768 if split else n * flop[0] / 4096) if flop else 0
        out[k] = {"avg_us": dur / 1e3, "effective_clock_GHz": ga / 8 / dur, "SQ_VALU_MFMA_BUSY_CYCLES": busy,
                  "products_per_multiply_add": products if split else 1, "expected_busy_cycles": (32 if split else 64) * n_mfma, "mfma_busy_fraction": busy / (ga / 8 * 1024),
                  "tflops": (n * flop[0] / (dur * 1e-9) / 1e12) if flop else None}
    print(json.dumps({"n_patches": n, "kernels": out,
                      "source": "rocprofv3 --pmc GRBM_GUI_ACTIVE / --pmc SQ_VALU_MFMA_BUSY_CYCLES (separate passes), tools/time_asdnet.py"}, indent=1))


if __name__ == "__main__":
    main()
