// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The 16-bit element of activations and weights in HBM / LDS and of the MFMA operands: bf16 in the default build
// (libgenpercept_hip.so: 8 significant bits, fp32 range), IEEE fp16 in the GP_F16 build (libgenpercept_hip_f16.so: 11 significant bits,
// range 65504 -- the reference's own half precision, run.py --half_precision).  v_mfma_f32_16x16x32_{bf16,f16} run at the same rate and
// use the same fragment layouts; accumulation, statistics, softmax and every epilogue computation are fp32 in both builds.  Everything
// format-specific lives in this block.
#ifndef GP_F16
#define GP_F16 0
#endif
typedef unsigned short h16_t;                                 // raw element bits
typedef short h16x8_t __attribute__((ext_vector_type(8)));    // MFMA A/B fragment (8 elements = 4 VGPR)
typedef float f32x4_t __attribute__((ext_vector_type(4)));    // 16x16 MFMA accumulator
typedef float f32x16_t __attribute__((ext_vector_type(16)));  // 32x32 MFMA accumulator
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#define GP_DEV __device__ __forceinline__

#if GP_F16
typedef _Float16 f16x2n_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4n_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8n_t __attribute__((ext_vector_type(8)));
GP_DEV float h16_to_f(h16_t x) { return (float)__builtin_bit_cast(_Float16, x); }
GP_DEV float h16_lo(unsigned u) { return (float)__builtin_bit_cast(f16x2n_t, u)[0]; }  // low element of a packed pair
GP_DEV float h16_hi(unsigned u) { return (float)__builtin_bit_cast(f16x2n_t, u)[1]; }  // high element of a packed pair
GP_DEV float h16_sat(float x) { return __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f); }  // fp16 has no headroom: saturate, never inf
// fp32 -> fp16, round-to-nearest-even, saturating (pack_*_ns: for values known to be in range -- probabilities, normalised inputs)
GP_DEV unsigned pack_h16x2_ns(float lo, float hi) {
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2n_t));
}
// Saturation is never silent: a saturating conversion that actually clips sets this TRANSLATION UNIT's flag word (one
// plain store on the clipping path only); the engine collects the flags of every translation unit into its counter at the end of each call
// (engine.hip: collect_saturation_kernel; gp_saturation_events; the pipeline logs a warning).  Hot epilogues carry
// the running max |value| of what they pack in a register (sat_track: one v_max3_f32 per pair, no branch) and report once per tile.
static __device__ unsigned gp_sat_flag;
GP_DEV float sat_track(float m, float a, float b) { return __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b))); }
GP_DEV void sat_report(float m) {
    if (m > 65504.f) *(volatile unsigned*)&gp_sat_flag = 1u;  // (NaN compares false: a NaN stays a NaN in the output, visible by itself)
}
GP_DEV unsigned pack_h16x2_t(float lo, float hi, float& m) {  // tracked: the caller reports m
    m = sat_track(m, lo, hi);
    return pack_h16x2_ns(h16_sat(lo), h16_sat(hi));
}
GP_DEV unsigned pack_h16x2(float lo, float hi) {
    sat_report(sat_track(0.f, lo, hi));
    return pack_h16x2_ns(h16_sat(lo), h16_sat(hi));
}
GP_DEV uint2 pack_h16x4_t(float a, float b, float c, float d, float& m) {
    m = sat_track(sat_track(m, a, b), c, d);
    f32x4_t v = {h16_sat(a), h16_sat(b), h16_sat(c), h16_sat(d)};
    return __builtin_bit_cast(uint2, __builtin_convertvector(v, f16x4n_t));
}
GP_DEV uint2 pack_h16x4(float a, float b, float c, float d) {
    float m = 0.f;
    const uint2 r = pack_h16x4_t(a, b, c, d, m);
    sat_report(m);
    return r;
}
#define GP_SAT_TU(name) void* gp_sat_flag_addr_##name() { void* q = nullptr; (void)hipGetSymbolAddress(&q, HIP_SYMBOL(gp_sat_flag)); return q; }
GP_DEV f32x4_t mfma_16x16x32(h16x8_t a, h16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8n_t, a), __builtin_bit_cast(f16x8n_t, b), c, 0, 0, 0);
}
GP_DEV f32x16_t mfma_32x32x16(h16x8_t a, h16x8_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8n_t, a), __builtin_bit_cast(f16x8n_t, b), c, 0, 0, 0);
}
#else
typedef __bf16 bf16x2n_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4n_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8n_t __attribute__((ext_vector_type(8)));
GP_DEV float h16_to_f(h16_t x) { return __uint_as_float(((unsigned)x) << 16); }
GP_DEV float h16_lo(unsigned u) { return __uint_as_float(u << 16); }          // low element of a packed pair
GP_DEV float h16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }  // high element of a packed pair
// fp32 -> bf16 round-to-nearest-even via v_cvt_pk_bf16_f32 (bf16 has the fp32 range: nothing to saturate)
GP_DEV unsigned pack_h16x2(float lo, float hi) {
    f32x2_t v = {lo, hi};
#ifdef GP_ROUND_ABL  // measurement build only (tools/build_round_abl.sh): one mantissa bit less, i.e. this translation unit's rounding error doubled
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2n_t)) & 0xfffefffeu;
#else
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2n_t));
#endif
}
GP_DEV unsigned pack_h16x2_ns(float lo, float hi) { return pack_h16x2(lo, hi); }
GP_DEV uint2 pack_h16x4(float a, float b, float c, float d) {
#ifdef GP_ROUND_ABL
    return make_uint2(pack_h16x2(a, b), pack_h16x2(c, d));
#else
    f32x4_t v = {a, b, c, d};
    return __builtin_bit_cast(uint2, __builtin_convertvector(v, bf16x4n_t));
#endif
}
// (bf16 has the fp32 range: the saturation tracking of the fp16 build compiles to nothing)
GP_DEV float sat_track(float m, float, float) { return m; }
GP_DEV void sat_report(float) {}
GP_DEV unsigned pack_h16x2_t(float lo, float hi, float&) { return pack_h16x2(lo, hi); }
GP_DEV uint2 pack_h16x4_t(float a, float b, float c, float d, float&) { return pack_h16x4(a, b, c, d); }
#define GP_SAT_TU(name) void* gp_sat_flag_addr_##name() { return nullptr; }
GP_DEV f32x4_t mfma_16x16x32(h16x8_t a, h16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8n_t, a), __builtin_bit_cast(bf16x8n_t, b), c, 0, 0, 0);
}
GP_DEV f32x16_t mfma_32x32x16(h16x8_t a, h16x8_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8n_t, a), __builtin_bit_cast(bf16x8n_t, b), c, 0, 0, 0);
}
#endif
GP_DEV h16_t f_to_h16(float f) { return (h16_t)(pack_h16x2(f, 0.f) & 0xffffu); }

GP_DEV float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }  // v_exp + v_rcp (1 ulp), no IEEE division sequence
// GELU(x) = x/2 (1 + erf(x / sqrt 2)), the exact (erf) form diffusers' GEGLU uses.  erf through Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7, far below the 16-bit output rounding): E = (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + p z), z = |x| / sqrt 2;
// 1 + erf = 2 - E for x >= 0 and = E for x < 0 (the complementary form: no cancellation in the negative tail).  About 16 VALU
// instructions with two transcendentals, no branches; the device library's erff is ~40 with two data-dependent paths.
GP_DEV float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float e = p * t * __builtin_amdgcn_exp2f(z * z * -1.44269504088896340736f);
    return 0.5f * x * (x >= 0.f ? 2.f - e : e);
}

// Asynchronous 16-byte-per-lane global -> LDS copy (LDS-DMA).  The LDS destination is wave-uniform base + lane*16,
// the global source is per lane; swizzles therefore go on the SOURCE address (cdna guide, rule 21).
GP_DEV void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
GP_DEV void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// The same LDS-DMA through a buffer resource (buffer_load_dwordx4 ... offen lds): base in four SGPRs, a 32-bit per-lane byte offset and
// a 32-bit uniform byte offset, no 64-bit address VGPRs.  Unlike global_load_lds (FLAT-encoded: while one is in flight hipcc's waitcnt
// pass degrades EVERY LDS wait to lgkmcnt(0)), the MUBUF form leaves the LDS counter alone, so ds_read results can be waited for one
// by one (counted lgkmcnt) underneath a DMA in flight.  Reads past `bytes` return zero.
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
GP_DEV buf_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
GP_DEV void blds16(buf_rsrc_t rsrc, unsigned lane_off, unsigned uniform_off, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, lane_off, uniform_off, 0, 0);
}

// LDS tiles are [rows][64 bf16] = 128-byte rows split into eight 16-byte slots.  Logical slot c of row r lives at
// physical slot c ^ (r & 7): with ds_read_b128's lane groups ({0-3,12-15,20-27}, ...) any window of sixteen consecutive
// rows read at slots (c, c+1) is conflict-free, whatever its first row (checked exhaustively; (r >> 1) & 7 is only
// conflict-free for windows starting at multiples of 4, which cost the halo conv 28 % of its LDS cycles).
GP_DEV int swz_slot(int row, int slot) { return slot ^ (row & 7); }
GP_DEV int lds_off128(int row, int slot) { return row * 128 + (swz_slot(row, slot) << 4); }

// XCD-aware bijective remap of a linear workgroup id: workgroup b runs on XCD b % 8 (observed), so give each XCD a
// contiguous chunk of the tile space to keep neighbouring tiles in one L2 (speed only, never correctness).
GP_DEV int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <int N>
GP_DEV void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// compile-time integer tag (generic lambdas instantiated per tap / ring slot / parity)
template <int V>
struct IC { static constexpr int value = V; };

// LDS accesses through INTEGER byte addresses: (a) a constant added to the address becomes the instruction's immediate offset,
// (b) hipcc does not treat them as possibly aliasing an LDS-DMA in flight (accesses it can trace to the `extern __shared__` array
// make it wait for vmcnt(0) first).  Ordering against the DMA ring is therefore entirely the kernel's business.
typedef const __attribute__((address_space(3))) h16x8_t* lds_frag_ptr;
typedef __attribute__((address_space(3))) f32x4_t* lds_f4_ptr;
typedef __attribute__((address_space(3))) float* lds_f_ptr;
GP_DEV h16x8_t lds_frag(unsigned base, int imm) { return *(lds_frag_ptr)(base + (unsigned)imm); }

// Sum over the 64 lanes of a wave, result uniform (an SGPR broadcast to every lane).  Pure VALU: four DPP adds inside each row of 16 lanes
// (xor 1, xor 2, half-mirror, mirror: after each step the paired groups already hold equal values, so any pairing of the groups works), two
// row-broadcast adds across the four rows, one v_readlane of lane 63.  `__shfl_xor` butterflies compile to ds_bpermute_b32 + s_waitcnt each:
// 216 serialized LDS round trips per four rows in cross_fold_kernel (r3: that chain, not memory, bound the kernel).
template <int CTRL, int ROW_MASK>
GP_DEV float dpp_move(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, true));
}
GP_DEV float wave_sum(float x) {
    x += dpp_move<0xB1, 0xf>(x);   // quad_perm [1,0,3,2]
    x += dpp_move<0x4E, 0xf>(x);   // quad_perm [2,3,0,1]
    x += dpp_move<0x141, 0xf>(x);  // row_half_mirror
    x += dpp_move<0x140, 0xf>(x);  // row_mirror: every lane of a row holds the row's sum
    x += dpp_move<0x142, 0xa>(x);  // row_bcast15 into rows 1 and 3
    x += dpp_move<0x143, 0xc>(x);  // row_bcast31 into rows 2 and 3: lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

// x[l] + x[l ^ 32] / x[l] + x[l ^ 16] in every lane: v_permlane32_swap / v_permlane16_swap (gfx950) exchange half-waves / odd-even rows of 16
// between two registers (semantics probed on the chip, tools/probe/permlane_probe.cpp: vdst' = [a.lo, b.lo], src' = [a.hi, b.hi]; rows
// [a0, b0, a2, b2] / [a1, b1, a3, b3]); with both registers holding x their sum is the pairwise total.  Through the BUILTINS, so that the
// compiler's hazard recognizer owns both sides (the wait states a VALU write needs before a permlane reads it, and whatever the
// consumer of the result needs): r3 issued them from inline asm with a hand-placed s_nop on the producer side only.
// (hipcc 7.2 miscompiles the direct form `r[0] + r[1]` of the builtin's result pair -- it emits v_add v1, v1, v1, dropping the second
//  register, also with distinct operands; seen in the ISA and as 81 failing GPU tests.  An empty asm over the two results pins them.)
#define GP_PERMLANE_PAIR(OP, x, r0, r1)                                             \
    const unsigned u_ = __builtin_bit_cast(unsigned, (x));                         \
    const auto rr_ = __builtin_amdgcn_##OP(u_, u_, false, false);                  \
    unsigned r0 = rr_[0], r1 = rr_[1];                                              \
    asm volatile("" : "+v"(r0), "+v"(r1))
GP_DEV float xor32_sum(float x) {
    GP_PERMLANE_PAIR(permlane32_swap, x, a, b);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
GP_DEV float xor16_sum(float x) {
    GP_PERMLANE_PAIR(permlane16_swap, x, a, b);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
GP_DEV float xor32_max(float x) {
    GP_PERMLANE_PAIR(permlane32_swap, x, a, b);
    return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
// sum over the lanes l' == l (mod SLW), SLW = 4, 8 or 16, valid in lanes 0 .. SLW-1 (every row's first SLW lanes): row_shl inside the rows of 16
// (lane i += lane i + 8, then + 4), then the two cross-row exchanges.  Replaces `for (off = 32; off >= SLW; off >>= 1) x += __shfl_xor(x, off)`,
// which is one ds_bpermute_b32 + s_waitcnt per value and step (the conv / GEMM epilogues reduce 16 statistics values per tile this way).
template <int SLW>
GP_DEV float slot_sum(float x) {
    static_assert(SLW == 4 || SLW == 8 || SLW == 16, "slot width");
    if (SLW <= 8) x += dpp_move<0x108, 0xf>(x);    // row_shl:8
    if (SLW == 4) x += dpp_move<0x104, 0xf>(x);    // row_shl:4
    return xor32_sum(xor16_sum(x));
}
// sum over each row of 16 lanes, valid in the row's lane 0
GP_DEV float row16_sum(float x) {
    x += dpp_move<0x108, 0xf>(x);
    x += dpp_move<0x104, 0xf>(x);
    x += dpp_move<0x102, 0xf>(x);
    x += dpp_move<0x101, 0xf>(x);
    return x;
}

// one group of the scheduler's instruction pattern (masks: 0x002 VALU, 0x008 MFMA, 0x100 DS read, 0x200 DS write, 0x400 transcendental)
template <int MASK, int N>
GP_DEV void sgb() { __builtin_amdgcn_sched_group_barrier(MASK, N, 0); }
