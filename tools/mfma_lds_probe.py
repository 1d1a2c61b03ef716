"""How much of an LDS fragment read overlaps with MFMA work on the same SIMD?  (gp_mfma_lds_probe, csrc/microbench.hip)

Prints one JSON object: TFLOP/s of the conv / GEMM inner loop in isolation -- 16 independent v_mfma_f32_16x16x32 plus R conflict-free
ds_read_b128 per wave and iteration, no barriers / DMA / epilogue -- for R in {0, 2, 4, 8, 16} at 1, 2 and 4 waves per SIMD, and next to it
what two simple models predict from the R = 0 rate: `overlap` (reads free until the LDS port saturates: 256 B/clk/CU) and `additive`
(every ds_read_b128 keeps its SIMD's matrix pipe idle for 16 cycles: 1 KiB returned at 64 B/clk).
    python tools/mfma_lds_probe.py [--precision bf16|fp16] > profiles/rNN_mfma_lds_probe.json
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args()
    from genpercept_amd import engine as ge
    rows = []
    for wps in (1, 2, 4):
        base = None
        for r in (0, 2, 4, 8, 16):
            tf = ge.mfma_lds_probe(a.device, r, wps, a.precision)
            if r == 0:
                base = tf
            # MFMA 16 cycles each on its SIMD; additive model: + 16 cycles per read on the same SIMD
            additive = base * 256.0 / (256.0 + 16.0 * r)
            # overlap model: the CU's LDS returns 256 B/clk: 4 SIMDs x r KiB per 256 MFMA cycles -> busy fraction r * 4 * 1024 / 256 / 256
            lds_busy = r * 4 * 1024 / 256.0 / 256.0
            overlap = base / max(1.0, lds_busy)
            rows.append({"waves_per_simd": wps, "reads_per_16_mfma": r, "tflops": round(tf, 1), "vs_no_reads": round(tf / base, 4),
                         "model_additive": round(additive, 1), "model_overlap": round(overlap, 1)})
    print(json.dumps({"probe": "16 x v_mfma_f32_16x16x32 + R x ds_read_b128 per wave and iteration, every CU, no barriers / DMA / epilogue",
                      "precision": a.precision, "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
