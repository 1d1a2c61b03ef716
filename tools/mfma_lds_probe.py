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
            tf = ge.mfma_lds_probe(a.device, r, wps, 0, a.precision)
            if r == 0:
                base = tf
            # MFMA 16 cycles each on its SIMD; additive model: + 16 cycles per read on the same SIMD
            additive = base * 256.0 / (256.0 + 16.0 * r)
            # overlap model: the CU's LDS returns 256 B/clk: 4 SIMDs x r KiB per 256 MFMA cycles -> busy fraction r * 4 * 1024 / 256 / 256
            lds_busy = r * 4 * 1024 / 256.0 / 256.0
            overlap = base / max(1.0, lds_busy)
            rows.append({"waves_per_simd": wps, "reads_per_16_mfma": r, "tflops": round(tf, 1), "vs_no_reads": round(tf / base, 4),
                         "model_additive": round(additive, 1), "model_overlap": round(overlap, 1)})
    # the conv kernels' operating point (8 reads per 16 MFMAs, two waves per SIMD) with their other ingredients added (gp_mfma_lds_probe mode bits)
    modes = [(8, 0, "inner loop alone"), (8, 1, "+ s_barrier per 32-MFMA step"), (8, 2, "+ weight stream (LDS-DMA ring, counted vmcnt)"),
             (8, 3, "+ barrier + weight stream"), (8, 7, "+ barrier + weight stream, weight fragments read from the ring"),
             (8, 11, "+ barrier + weight stream as buffer_load ... lds"), (8, 15, "the same, fragments from the ring"),
             (8, 19, "+ barrier + weight stream + halo stream (48 KiB / 9 steps from HBM)"), (8, 23, "the same, fragments from the ring"),
             (0, 0, "no fragment reads at all"), (0, 1, "no reads, + barrier"), (0, 3, "no reads, + barrier + weight stream"),
             (0, 19, "no reads, + barrier + weight stream + halo stream")]
    steps = []
    for rep in range(2):
        for r, m, what in modes:
            tf = ge.mfma_lds_probe(a.device, r, 2, m, a.precision)
            if rep == 0:
                steps.append({"reads_per_16_mfma": r, "mode": m, "what": what, "tflops": [round(tf, 1)]})
            else:
                [x for x in steps if x["reads_per_16_mfma"] == r and x["mode"] == m][0]["tflops"].append(round(tf, 1))
    print(json.dumps({"build_up_at_2_waves_per_simd": steps, "probe": "16 x v_mfma_f32_16x16x32 + R x ds_read_b128 per wave and iteration, every CU, no barriers / DMA / epilogue",
                      "precision": a.precision, "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
