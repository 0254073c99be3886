// hjd_pixels.hip — block back end (gfx950): dequantise, 8x8 IDCT, chroma upsample, YCbCr->RGB,
// output layout.
//   k_idct_color   DecodeSingleBlock + YCrCB_to_RGB24_Block8x8 (+ WriteBMP24 layout)
//                  (loadjpg.cpp:184-228, :867-932; openjpg.cpp:504-570)
//   k_color_table  test hook: the colour stage alone over all 2^24 (Y, Cb, Cr) triples
//
// Compile with -ffp-contract=off: the reference's IDCT and colour arithmetic are fp32 with
// one rounding per operation; an FMA would change output bytes.
//
// One workgroup = up to 16 MCUs of one MCU row.  What bounds it on MI355X is not HBM but how long a
// workgroup lives: round 1's version spent 41 k cycles per workgroup, most of them on nine dependent
// global round trips (image search, image fields, offsets, records) and eight barriers, for 6 k cycles of
// own instruction issue.  This version: workgroup -> (image, first block) is one table load, the block
// entries and the image fields are fetched together, records are scattered by a flat, fully coalesced
// pass (each record carries its block and its slot), the slot masks come ready from k_write, blocks are
// dealt to the 4-lane teams component-major (64 luma blocks, then the chroma blocks: similar loop
// lengths without a sort), and three barriers remain.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hjd_device.h"
#include "hjd_kernels.h"
#include "hjd_kutil.h"

// COS[x][u] = cosf(fl((2x+1)*u*3.14f)/16) exactly as the reference's x86-64/glibc-2.35 build
// evaluates it (loadjpg.cpp:108,120).  A table, never recomputed: entry [6][7]
// is libm-specific, and PI = 3.14f makes the basis non-orthogonal, so no fast IDCT is bit-exact.
__constant__ float c_cos[64] = {
    0x1p+0f, 0x1.f62c08p-1f, 0x1.d910b8p-1f, 0x1.a9cc2p-1f, 0x1.6a2ecap-1f, 0x1.1ca9eep-1f, 0x1.886ecap-2f, 0x1.90f1bep-3f,
    0x1p+0f, 0x1.a9cc2p-1f, 0x1.886ecap-2f, -0x1.8dbedp-3f, -0x1.699b2p-1f, -0x1.f6032cp-1f, -0x1.d9606ap-1f, -0x1.1d5756p-1f,
    0x1p+0f, 0x1.1ca9eep-1f, -0x1.86ed1p-2f, -0x1.f6032cp-1f, -0x1.6ac236p-1f, 0x1.8a8b92p-3f, 0x1.d8706ep-1f, 0x1.aab36ap-1f,
    0x1p+0f, 0x1.90f1bep-3f, -0x1.d8c0bap-1f, -0x1.1d5756p-1f, 0x1.690742p-1f, 0x1.aab36ap-1f, -0x1.83e8d8p-2f, -0x1.f6a4aep-1f,
    0x1p+0f, -0x1.8dbedp-3f, -0x1.d9606ap-1f, 0x1.1b4e94p-1f, 0x1.6b556cp-1f, -0x1.a86f2p-1f, -0x1.8cf256p-2f, 0x1.f5869ep-1f,
    0x1p+0f, -0x1.1bfc56p-1f, -0x1.89f044p-2f, 0x1.f67ccap-1f, -0x1.68731cp-1f, -0x1.9a88c4p-3f, 0x1.da4da2p-1f, -0x1.a7851ep-1f,
    0x1p+0f, -0x1.a9581p-1f, 0x1.856b06p-2f, 0x1.9756cp-3f, -0x1.6be866p-1f, 0x1.f6cc3cp-1f, -0x1.d77dbp-1f, 0x1.1895aep-1f,
    0x1p+0f, -0x1.f6032cp-1f, 0x1.d8706ep-1f, -0x1.a86f2p-1f, 0x1.67dec4p-1f, -0x1.19442cp-1f, 0x1.7f609p-2f, -0x1.7a87f6p-3f};

// Conformant mode only (hjd_options.conformant, SURVEY §8f N4 — never the default): COS[x][u] = cos((2x+1) u pi / 16).
__constant__ float c_cos_true[64] = {
    1.0f, 0.98078528f, 0.92387953f, 0.83146961f, 0.70710678f, 0.55557023f, 0.38268343f, 0.19509032f,
    1.0f, 0.83146961f, 0.38268343f, -0.19509032f, -0.70710678f, -0.98078528f, -0.92387953f, -0.55557023f,
    1.0f, 0.55557023f, -0.38268343f, -0.98078528f, -0.70710678f, 0.19509032f, 0.92387953f, 0.83146961f,
    1.0f, 0.19509032f, -0.92387953f, -0.55557023f, 0.70710678f, 0.83146961f, -0.38268343f, -0.98078528f,
    1.0f, -0.19509032f, -0.92387953f, 0.55557023f, 0.70710678f, -0.83146961f, -0.38268343f, 0.98078528f,
    1.0f, -0.55557023f, -0.38268343f, 0.98078528f, -0.70710678f, -0.19509032f, 0.92387953f, -0.83146961f,
    1.0f, -0.83146961f, 0.38268343f, 0.19509032f, -0.70710678f, 0.98078528f, -0.92387953f, 0.55557023f,
    1.0f, -0.98078528f, 0.92387953f, -0.83146961f, 0.70710678f, -0.55557023f, 0.38268343f, -0.19509032f};

#define MAXGB (HJD_GROUP_MCUS * 6) // blocks per workgroup
#define HJD_PIX_ROW 132            // bytes between the 8x8 tiles of two blocks in LDS: a tile takes the place of its block's coefficients

// ---- colour ------------------------------------------------------------------------------------
// ConvertYCrCbtoRGB as called at loadjpg.cpp:918 (:867-880): R = (int)(Y + 1.402 (Cr-128)),
// G = (int)((Y - 0.34414 (Cb-128)) - 0.71414 (Cr-128)), B = (int)(Y + 1.772 (Cb-128)), one fp32
// rounding per operation, then Clamp (loadjpg.cpp:83-91).  (int) truncates toward zero and Clamp
// saturates to 0..255; v_trunc_f32 followed by v_cvt_pk_u8_f32 (float -> u8 with saturation, written
// into a chosen byte of a dword) gives the same byte for every finite input: truncation first makes
// the value integral, so the conversion's own rounding never acts, negative values saturate to 0
// like Clamp((int)x) does.  (Without the v_trunc_f32 the conversion rounds to nearest and the exhaustive
// device test fails — tried.)  k_color_table runs these functions over all 2^24 inputs the colour stage
// can ever see (tests/test_gpu_parity.py::test_colour_stage_exhaustive).

struct ChromaTerms {
    float a, b, c, d; // 1.402 cr', 0.34414 cb', 0.71414 cr', 1.772 cb'   (x' = x - 128)
};

template <bool CONF = false>
__device__ __forceinline__ ChromaTerms chroma_terms(uint32_t cb, uint32_t cr)
{
    const float fcb = (float)cb - 128.0f, fcr = (float)cr - 128.0f; // == (float)(c - 128) exactly
    ChromaTerms t;
    t.a = 1.402f * fcr;
    t.b = (CONF ? 0.344136f : 0.34414f) * fcb; // conformant mode: the JFIF constants
    t.c = (CONF ? 0.714136f : 0.71414f) * fcr;
    t.d = 1.772f * fcb;
    return t;
}

// byte `sel` of `acc` <- clamp((int)x, 0, 255)
template <bool CONF = false>
__device__ __forceinline__ uint32_t put_u8(float x, uint32_t sel, uint32_t acc)
{
    return __builtin_amdgcn_cvt_pk_u8_f32(CONF ? __builtin_floorf(x + 0.5f) : __builtin_truncf(x), sel, acc); // conformant: round
}

// Four pixels -> the 12 bytes c0,g,c2 x 4 (c0 = R and c2 = B for the m_rgb layout, swapped for BMP).
// y4 = four luma bytes; t0..t3 = chroma terms of the four pixels (4:2:x callers pass each pair twice).
template <bool bgr, bool CONF = false>
__device__ __forceinline__ void rgb12(uint32_t y4, const ChromaTerms &t0, const ChromaTerms &t1, const ChromaTerms &t2,
                                      const ChromaTerms &t3, uint32_t &d0, uint32_t &d1, uint32_t &d2)
{
    const float y0 = (float)(y4 & 0xFF), y1 = (float)((y4 >> 8) & 0xFF), y2 = (float)((y4 >> 16) & 0xFF), y3 = (float)(y4 >> 24);
    const float r0 = y0 + t0.a, g0 = (y0 - t0.b) - t0.c, b0 = y0 + t0.d;
    const float r1 = y1 + t1.a, g1 = (y1 - t1.b) - t1.c, b1 = y1 + t1.d;
    const float r2 = y2 + t2.a, g2 = (y2 - t2.b) - t2.c, b2 = y2 + t2.d;
    const float r3 = y3 + t3.a, g3 = (y3 - t3.b) - t3.c, b3 = y3 + t3.d;
    const float f0 = bgr ? b0 : r0, l0 = bgr ? r0 : b0;
    const float f1 = bgr ? b1 : r1, l1 = bgr ? r1 : b1;
    const float f2 = bgr ? b2 : r2, l2 = bgr ? r2 : b2;
    const float f3 = bgr ? b3 : r3, l3 = bgr ? r3 : b3;
    if (CONF) {
        d0 = put_u8<CONF>(f1, 3, put_u8<CONF>(l0, 2, put_u8<CONF>(g0, 1, put_u8<CONF>(f0, 0, 0u))));
        d1 = put_u8<CONF>(g2, 3, put_u8<CONF>(f2, 2, put_u8<CONF>(l1, 1, put_u8<CONF>(g1, 0, 0u))));
        d2 = put_u8<CONF>(l3, 3, put_u8<CONF>(g3, 2, put_u8<CONF>(f3, 1, put_u8<CONF>(l2, 0, 0u))));
    } else {
        // (int) truncates, v_cvt_pk_u8_f32 converts in the wave's fp32 rounding mode: switch that to round-toward-zero for
        // the twelve conversions (one statement, so that no other arithmetic can be scheduled inside) instead of a
        // v_trunc_f32 in front of each.  Negative values still saturate to 0, like Clamp((int)x).
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\ts_nop 1\n\t"
                     "v_cvt_pk_u8_f32 %0, %3, 0, 0\n\tv_cvt_pk_u8_f32 %0, %4, 1, %0\n\tv_cvt_pk_u8_f32 %0, %5, 2, %0\n\tv_cvt_pk_u8_f32 %0, %6, 3, %0\n\t"
                     "v_cvt_pk_u8_f32 %1, %7, 0, 0\n\tv_cvt_pk_u8_f32 %1, %8, 1, %1\n\tv_cvt_pk_u8_f32 %1, %9, 2, %1\n\tv_cvt_pk_u8_f32 %1, %10, 3, %1\n\t"
                     "v_cvt_pk_u8_f32 %2, %11, 0, 0\n\tv_cvt_pk_u8_f32 %2, %12, 1, %2\n\tv_cvt_pk_u8_f32 %2, %13, 2, %2\n\tv_cvt_pk_u8_f32 %2, %14, 3, %2\n\t"
                     "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\ts_nop 1"
                     : "=&v"(d0), "=&v"(d1), "=&v"(d2)
                     : "v"(f0), "v"(g0), "v"(l0), "v"(f1), "v"(g1), "v"(l1), "v"(f2), "v"(g2), "v"(l2), "v"(f3), "v"(g3), "v"(l3));
    }
}

// Phase 3 of k_idct_color: nearest-neighbour chroma (loadjpg.cpp:911-912), colour, clip at the image edge
// (:906-909), then either the m_rgb layout or the WriteBMP24 layout (openjpg.cpp:555-568).  A thread owns one
// 4-pixel column of the group for good and walks chroma rows; everything that depends on the column only is
// computed once, and the chroma products once per chroma sample row (they serve vf luma rows).  HF (1, 2 or 4,
// hjd_parse_header) is a template parameter: the unit grid, the chroma fetch and the term sharing are then fixed
// code; the output pointer is stepped by the row stride instead of being rebuilt, and the 12-byte store's
// alignment is decided once per thread whenever the stride is a multiple of 4.
template <bool BMP, bool CONF, uint32_t HF>
__device__ __forceinline__ void color_store_hf(const HjdImage &im, uint8_t *__restrict__ out, const uint8_t (*s_pix)[HJD_PIX_ROW], uint32_t tid,
                                               uint32_t mrow, uint32_t mcu_first, uint32_t K, uint32_t lg)
{
    const uint32_t W = im.width, H = im.height, vf = im.vf, nb = im.nb, nby = im.nby;
    const uint32_t stride = im.out_stride;
    constexpr uint32_t LHF = HF >> 1;      // log2 for 1, 2, 4
    constexpr uint32_t LUPR = 5u + LHF;    // log2(4-pixel units per padded row) = log2(HJD_GROUP_MCUS * 2 * HF)
    constexpr uint32_t MW = 8u * HF;
    const uint32_t mh = 8u * vf, lvf = vf >> 1;
    // The unit grid is padded to HJD_GROUP_MCUS MCUs per row so that a unit's row and column are bit fields of the thread index.
    const uint32_t ux = tid & ((1u << LUPR) - 1u);
    const uint32_t px = ux * 4u;
    const uint32_t k = px >> (3u + LHF), xin = px & (MW - 1u);
    const uint32_t gxp = (mcu_first + k) * MW + xin;
    if (ux < K * 2u * HF && gxp < W) { // loadjpg.cpp:907
        const uint32_t nvalid = min(4u, W - gxp);
        const uint8_t *ytile = &s_pix[k * nb + (xin >> 3)][xin & 7u]; // luma tile of this column in the MCU's first tile row
        const uint8_t *pcb0 = &s_pix[k * nb + nby][0] + (HF == 2 ? xin >> 1 : (HF == 4 ? xin >> 2 : xin));
        const uint8_t *pcr0 = pcb0 + sizeof(s_pix[0]);
        const uint32_t gy0 = mrow * mh;
        // output address of (gxp, gy0) and the step to the next image row (BMP: bottom-up, openjpg.cpp:555-568)
        uint8_t *const dst0 = BMP ? out + 54 + (size_t)(H - 1u - gy0) * stride + (size_t)gxp * 3u : out + (size_t)gy0 * stride + (size_t)gxp * 3u;
        const ptrdiff_t rstep = BMP ? -(ptrdiff_t)stride : (ptrdiff_t)stride;
        const bool wide = nvalid == 4u && (stride & 3u) == 0u && (reinterpret_cast<uintptr_t>(dst0) & 3u) == 0u; // every row's 12 bytes are aligned
        for (uint32_t cy = tid >> LUPR; cy < 8u; cy += 256u >> LUPR) {
            ChromaTerms t0, t1, t2, t3;
            if (HF == 2) { // pixels 0,1 and 2,3 share a chroma sample (cx = x/2)
                const uint32_t cb2 = *reinterpret_cast<const uint16_t *>(&pcb0[cy * 8u]);
                const uint32_t cr2 = *reinterpret_cast<const uint16_t *>(&pcr0[cy * 8u]);
                t0 = t1 = chroma_terms<CONF>(cb2 & 0xFFu, cr2 & 0xFFu);
                t2 = t3 = chroma_terms<CONF>(cb2 >> 8, cr2 >> 8);
            } else if (HF == 4) {
                t0 = t1 = t2 = t3 = chroma_terms<CONF>(pcb0[cy * 8u], pcr0[cy * 8u]);
            } else {
                const uint32_t cb4 = *reinterpret_cast<const uint32_t *>(&pcb0[cy * 8u]);
                const uint32_t cr4 = *reinterpret_cast<const uint32_t *>(&pcr0[cy * 8u]);
                t0 = chroma_terms<CONF>(cb4 & 0xFFu, cr4 & 0xFFu);
                t1 = chroma_terms<CONF>((cb4 >> 8) & 0xFFu, (cr4 >> 8) & 0xFFu);
                t2 = chroma_terms<CONF>((cb4 >> 16) & 0xFFu, (cr4 >> 16) & 0xFFu);
                t3 = chroma_terms<CONF>(cb4 >> 24, cr4 >> 24);
            }
            const uint32_t row0 = cy << lvf;
            uint8_t *dst = dst0 + (ptrdiff_t)row0 * rstep;
            // the vf luma rows of one chroma row lie in the same tile row (vf divides 8): one address, stepped by a tile line
            const uint8_t *yp = ytile + (row0 >> 3) * (HF * (uint32_t)sizeof(s_pix[0])) + (row0 & 7u) * 8u;
            for (uint32_t dy = 0; dy < vf; dy++, dst += rstep, yp += 8) {
                const uint32_t row = row0 + dy;
                if (gy0 + row >= H) break; // loadjpg.cpp:908
                const uint32_t ypix = *reinterpret_cast<const uint32_t *>(yp);
                uint32_t d0, d1, d2;
                rgb12<BMP, CONF>(ypix, t0, t1, t2, t3, d0, d1, d2);
                if (wide || (nvalid == 4u && (reinterpret_cast<uintptr_t>(dst) & 3u) == 0u)) {
                    // one 12-byte store: a wave writes 768 contiguous bytes.  Non-temporal: nobody on the GPU reads the pixels back,
                    // and 25 MB per image streamed through the L2 pushed out what the other kernels of the step were about to
                    // use (the record array, half-written sectors of block entries): 284 -> 293 Gpixel/s, A/B on one box
                    typedef uint32_t px12_t __attribute__((ext_vector_type(3)));
                    __builtin_nontemporal_store((px12_t){d0, d1, d2}, reinterpret_cast<px12_t __attribute__((aligned(4))) *>(dst));
                } else {
                    const uint32_t dd[3] = {d0, d1, d2};
                    for (uint32_t j = 0; j < nvalid * 3u; j++) dst[j] = (uint8_t)(dd[j >> 2] >> ((j & 3u) * 8u));
                }
                if (BMP && gxp + nvalid == W) // row padding (openjpg.cpp:563-567), written as zeros
                    for (uint32_t q = 3u * W; q < stride; q++) dst[(ptrdiff_t)q - (ptrdiff_t)(gxp * 3u)] = 0;
            }
        }
    }
    if (BMP && lg == 0 && tid < 54) {
        // stBMFH (openjpg.cpp:507-547): 'BM', file size, 0, 0, 54, 40, W, H, planes 1, 24 bpp, rest 0
        const uint32_t fsz = W * H * 3 + H * (stride - 3 * W) + 54;
        uint32_t v = 0;
        if (tid == 0) v = 'B';
        else if (tid == 1) v = 'M';
        else if (tid >= 2 && tid < 6) v = (fsz >> (8 * (tid - 2))) & 0xFF;
        else if (tid == 10) v = 54;
        else if (tid == 14) v = 40;
        else if (tid >= 18 && tid < 22) v = (W >> (8 * (tid - 18))) & 0xFF;
        else if (tid >= 22 && tid < 26) v = (H >> (8 * (tid - 22))) & 0xFF;
        else if (tid == 26) v = 1;
        else if (tid == 28) v = 24;
        out[tid] = (uint8_t)v;
    }
}

template <bool BMP, bool CONF>
__device__ __forceinline__ void color_store(const HjdImage &im, uint8_t *__restrict__ out, const uint8_t (*s_pix)[HJD_PIX_ROW], uint32_t tid,
                                            uint32_t mrow, uint32_t mcu_first, uint32_t K, uint32_t lg)
{
    if (im.hf == 2) color_store_hf<BMP, CONF, 2u>(im, out, s_pix, tid, mrow, mcu_first, K, lg);
    else if (im.hf == 1) color_store_hf<BMP, CONF, 1u>(im, out, s_pix, tid, mrow, mcu_first, K, lg);
    else color_store_hf<BMP, CONF, 4u>(im, out, s_pix, tid, mrow, mcu_first, K, lg);
}

// ---- stage 4: dequantise + IDCT + upsample + colour + store -----------------------------------

// CONF: conformant mode (hjd_options.conformant) — a separate instantiation, so that the default kernel carries none of
// its branches (as run-time branches they cost 4 % of the kernel); every image of a batch has the decoder's setting.
#ifdef HJD_IDCT_WAVES // experiment: cap the resident waves per SIMD so that other streams' large-LDS workgroups find room beside this kernel
#define HJD_IDCT_OCC __attribute__((amdgpu_waves_per_eu(HJD_IDCT_WAVES, HJD_IDCT_WAVES)))
#else
#define HJD_IDCT_OCC
#endif
template <bool CONF>
__global__ __launch_bounds__(256) HJD_IDCT_OCC void k_idct_color(HjdParams P)
{
    __shared__ __attribute__((aligned(8))) int16_t s_coef[MAXGB][66]; // dequantised, index = slot u*8+v (u = horizontal frequency); +2 pad: bank spread
    __shared__ uint32_t s_mask[MAXGB][2]; // bit = slot holds a coefficient (lo: slots 0-31, hi: 32-63)
    __shared__ uint8_t s_comp[MAXGB + 32]; // component (0 Y, 1 Cb, 2 Cr) of every block of the group
    // The decoded 8x8 tile of a block (row-major, 64 bytes) is written over the block's coefficients: only the block's own
    // 4-lane team reads them, and it has read them all when it stores the first pixel.  (With an array of their own the tiles
    // took 6.9 KB more: seven workgroups per CU instead of eight.)
    static_assert(sizeof(int16_t) * 66 == HJD_PIX_ROW, "one tile per coefficient row");
    uint8_t (*s_pix)[HJD_PIX_ROW] = reinterpret_cast<uint8_t (*)[HJD_PIX_ROW]>(&s_coef[0][0]);
    __shared__ __attribute__((aligned(16))) float s_cos[64]; // [u][x] = COS[x][u]; read as [v][y] = COS[y][v] too
    __shared__ float s_kk[64];            // C(u)*C(v) by slot
    __shared__ __attribute__((aligned(4))) uint8_t s_qt[3][64]; // by slot
    __shared__ uint32_t s_range[2];       // records of the group: [first, end)
    __shared__ uint32_t s_hist[64];       // counting sort of the blocks by number of terms: bin fill, then bin start
    __shared__ uint8_t s_order[MAXGB];    // blocks in descending order of their term count

    const uint32_t tid = threadIdx.x;
    // second launch, after the sequential emulation (gate_emulated): its grid is the list of the groups of the images that carry DRI,
    // and of those only the images the emulation has just decoded go on
    const uint32_t bid = P.gate_emulated ? P.rst_groups[blockIdx.x] : blockIdx.x + P.off_group;
    const uint2 gm = P.map_group[bid];
    const int ii = (int)gm.x;
    if (P.gate_emulated && !P.st[ii].emulated) return;
    const HjdImage &im = P.img[ii];
    const uint32_t lg = bid - im.group0;
    const uint32_t mrow = lg / im.groups_per_row, gx = lg % im.groups_per_row;
    const uint32_t mcu_first = gx * HJD_GROUP_MCUS;
    const uint32_t K = min((uint32_t)HJD_GROUP_MCUS, im.mcux - mcu_first);
    const uint32_t nb = im.nb, nby = im.nby;
    const uint32_t KB = K * nb;

    // the group's record range (first record of its first block, first record of the block after its last) and the predicted
    // DC values — independent of the image fields
    uint32_t bfirst = 0u;
    int dc = 0;
    if (tid == 0 || tid == KB) bfirst = P.blk[gm.y + tid].x;
    if (tid < KB) dc = P.dca[gm.y + tid];
    constexpr bool conf = CONF;
    if (tid < 64) {
        s_cos[tid] = conf ? c_cos_true[(tid & 7) * 8 + (tid >> 3)] : c_cos[(tid & 7) * 8 + (tid >> 3)];
        // C(u)*C(v): fl(C0*C0) = 0x1.fffffep-2 (not 0.5), C0 = 1/sqrtf(2) = 0x1.6a09e6p-1 (loadjpg.cpp:96-102,120)
        const uint32_t u = tid >> 3, v = tid & 7;
        // ... times the 0.25 of loadjpg.cpp:123: scaling by a power of two commutes with every fp32 rounding of the sum (no
        // term comes near the denormals), so the accumulators hold 0.25 * sum exactly and the final multiplication is gone
        s_kk[tid] = 0.25f * ((u == 0) ? (v == 0 ? (conf ? 0.5f : 0x1.fffffep-2f) : 0x1.6a09e6p-1f) : (v == 0 ? 0x1.6a09e6p-1f : 1.0f));
    }
    if (tid < 48) reinterpret_cast<uint32_t *>(&s_qt[0][0])[tid] = reinterpret_cast<const uint32_t *>(&im.qt[0][0])[tid];
    if (tid >= 64 && tid < 128) s_hist[tid - 64] = 0u;
    if (tid == 0) s_range[0] = min(bfirst, im.rec_cap);
    if (tid == KB) s_range[1] = min(bfirst, im.rec_cap);
    uint32_t mycomp = 0;
    if (tid < KB) {
        // tid % nb without a division: nb is 3, 4 or 6 (hf*vf <= 4) and tid < 256, where floor(tid/nb) == tid * ceil(2^16/nb) >> 16
        const uint32_t inv = nb == 4u ? 16384u : (nb == 3u ? 21846u : 10923u);
        const uint32_t mm = tid - ((tid * inv) >> 16) * nb;
        mycomp = mm < nby ? 0u : mm - nby + 1u;
        s_comp[tid] = (uint8_t)mycomp;
        s_mask[tid][0] = 0u; // filled from the records below
        s_mask[tid][1] = 0u;
    }
    __syncthreads();
    if (tid < KB) { // DC: predicted value from k_dcapply, quantiser slot 0 (DequantizeBlock, loadjpg.cpp:144-152: int16 wrap)
        if (conf && im.ri_blocks) {
            // conformant mode: predictors restart with every interval = the running sum minus its value at the
            // end of the previous interval (the reference, and the default mode, carry it across: loadjpg.cpp:1159-1162)
            const uint32_t ri = im.ri_blocks / nb, mcu = (uint32_t)(gm.y - im.blk_off + tid) / nb;
            const uint32_t s0 = mcu - mcu % ri;
            if (s0) dc = (int)(int16_t)(dc - (int)P.dca[im.blk_off + (size_t)(s0 - 1u) * nb + (mycomp == 0 ? nby - 1u : nby + mycomp - 1u)]);
        }
        const int16_t nat = (int16_t)(dc * (int)s_qt[mycomp][0]);
        s_coef[tid][0] = nat;
        if (nat != 0) atomicOr(&s_mask[tid][0], 1u);
    }

    // phase 1: DequantizeBlock on the sparse form.  The group's records are one contiguous run (padding
    // included); every record names its block (low 7 bits of the block index) and its slot, so the run is
    // walked flat, 16 bytes per lane, fully coalesced.  int16 wrap of the product as the reference's
    // (short)(int) cast.
    {
        const uint32_t *rec = P.rec + im.rec_off;
        // A group holds at most 63 records per block, and every k_write lane's run with k >= 1 of them inside the group
        // is padded to at most k + HJD_REC_RUN - 1 <= HJD_REC_RUN * k entries: that bounds the span of any real group
        // (round 2 capped it at MAXGB * 64 + 1024, which dense q100 content exceeds at 64-bit subsequences: records
        // dropped without a status).  Entries left over from a run that failed half-way (the image then carries an error
        // status) still cannot turn this loop into a walk over the whole record array.
        const uint32_t r0 = s_range[0];
        const uint32_t r1 = min(s_range[1], r0 + HJD_REC_RUN * 63u * KB + HJD_REC_RUN);
        const uint32_t b0 = (uint32_t)(gm.y - im.blk_off);
        for (uint32_t t = (r0 & ~3u) + tid * 4u; t < r1; t += 1024u) {
            typedef uint32_t rec4_t __attribute__((ext_vector_type(4)));
            const rec4_t q = __builtin_nontemporal_load(reinterpret_cast<const rec4_t *>(rec + t)); // read once: not worth a place in the L2 (+2 %)
            const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t w = w4[k];
                const uint32_t rel = ((w >> 22) - b0) & 127u;
                if (w != HJD_REC_PAD && t + k >= r0 && t + k < r1 && rel < KB) {
                    const uint32_t slot = (w >> 16) & 63u;
                    s_coef[rel][slot] = (int16_t)((int)(int16_t)(w & 0xFFFFu) * (int)s_qt[s_comp[rel]][slot]);
                    atomicOr(&s_mask[rel][slot >> 5], 1u << (slot & 31u)); // the slots a block's records fill: its term list
                }
            }
        }
    }
    __syncthreads();
#if defined(HJD_IDCT_CUT) && HJD_IDCT_CUT == 1 // instruction accounting only (tools/idct_phase_counts.sh): stop after phase 1
    P.out[im.out_off + (size_t)lg * 256u + tid] = (uint8_t)s_coef[tid % MAXGB][tid & 63];
    return;
#endif

    // The 16 teams of a wave walk their blocks' terms in lock step, so a wave pays for its LONGEST block: blocks are
    // dealt to the teams in descending order of their term count (counting sort over the group: bin = 63 - key,
    // key = max(terms in slots 1-31, 4 x terms in slots 32-63), the two loops below).  On the 4K synthetic set this
    // takes the term loop from 13.1 to 11.0 iterations per wave pass (tools/idct_model.py).
    uint32_t bin = 0, rank = 0;
    if (tid < KB) {
        const uint2 mk = *reinterpret_cast<const uint2 *>(&s_mask[tid][0]);
        bin = 63u - min(63u, max((uint32_t)__popc(mk.x & ~1u), 4u * (uint32_t)__popc(mk.y)));
        rank = atomicAdd(&s_hist[bin], 1u);
    }
    __syncthreads();
    if (tid < 64) { // exclusive prefix over the 64 bins, one wave
        const uint32_t c = s_hist[tid];
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d, 64);
            if ((int)tid >= d) incl += o;
        }
        s_hist[tid] = incl - c;
    }
    __syncthreads();
    if (tid < KB) s_order[s_hist[bin] + rank] = (uint8_t)tid;
    __syncthreads();

    // phase 2: IDCT_calc / PerformIDCT (loadjpg.cpp:105-140).  Team of 4 lanes = one block; lane = two output
    // columns x0, x0+1 with 2 x 8 row accumulators.  Terms are added in the reference's order (u outer, v
    // inner = ascending slot); empty slots would add +-0 and are skipped.
    // term = (((C(u)C(v)) * b) * COS[x][u]) * COS[y][v].
    {
        const uint32_t x0 = (tid & 3) * 2;
        // sorted positions 0..63 go to the four waves in order; positions 64.. (the shortest blocks) to waves 3, 2 — the
        // waves whose first blocks were the shortest
        for (uint32_t pass = 0; pass < 2; pass++) {
            const uint32_t s = pass == 0 ? (tid >> 2) : 64u + (3u - (tid >> 6)) * 16u + ((tid >> 2) & 15u);
            if (s >= KB) continue;
            const uint32_t blk = s_order[s];
            const uint2 mk = *reinterpret_cast<const uint2 *>(&s_mask[blk][0]);
            const int16_t *cf = &s_coef[blk][0];
            // The first term of the sum is the DC term whenever slot 0 is filled: COS[x][0] = COS[y][0] = 1.0f, both
            // multiplications are exact and 0.0f + t = t, so the accumulators simply START at C(0)C(0)*DC — one
            // whole pass of the term loop less for every block (all 16 teams of a wave have it).
            const float tdc = (mk.x & 1u) ? s_kk[0] * (float)cf[0] : 0.0f;
            float acc0[8], acc1[8];
#pragma unroll
            for (int y = 0; y < 8; y++) acc0[y] = acc1[y] = tdc;
#pragma unroll
            for (int half = 0; half < 2; half++) { // u = 0..3, then u = 4..7: ascending set bits = the term order
                uint32_t m32 = half ? mk.y : (mk.x & ~1u);
                while (m32) {
                    const uint32_t j = half * 32u + (uint32_t)__ffs((int)m32) - 1u;
                    m32 &= m32 - 1u;
                    const uint32_t u = j >> 3, v = j & 7;
                    const float t1 = s_kk[j] * (float)cf[j];
                    const float2 cx = *reinterpret_cast<const float2 *>(&s_cos[u * 8 + x0]);
                    const float t2a = t1 * cx.x, t2b = t1 * cx.y;
                    const float4 ca = *reinterpret_cast<const float4 *>(&s_cos[v * 8]);
                    const float4 cb = *reinterpret_cast<const float4 *>(&s_cos[v * 8 + 4]);
                    const float cy[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
#pragma unroll
                    for (int y = 0; y < 8; y++) {
                        acc0[y] = acc0[y] + t2a * cy[y];
                        acc1[y] = acc1[y] + t2b * cy[y];
                    }
                }
            }
            // (int)(0.25*sum) — the accumulators hold 0.25*sum — then short += 128, then Clamp (loadjpg.cpp:123,136-137,219).  The short only wraps
            // for |0.25*sum| >= 32640; below that Clamp((short)((int)t + 128)) == saturate_u8(trunc(t) + 128),
            // which is what v_cvt_pk_u8_f32 computes from the integral float trunc(t) + 128 (exact below 2^23).
            float big = 0.0f; // max |.| over the 16 values: one v_max3_f32 with |.| source modifiers per pair (the compiler spends
                              // three instructions on fmaxf(fabsf(a), fabsf(b)))
#pragma unroll
            for (int y = 0; y < 8; y++) asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(big) : "v"(acc0[y]), "v"(acc1[y]));
            if (conf) { // conformant mode: round to nearest
#pragma unroll
                for (int y = 0; y < 8; y++) {
                    const uint32_t pp = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(acc1[y] + 128.5f), 1,
                                        __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(acc0[y] + 128.5f), 0, 0u));
                    *reinterpret_cast<uint16_t *>(&s_pix[blk][y * 8 + x0]) = (uint16_t)pp;
                }
            } else if (big < 32000.0f) {
#pragma unroll
                for (int y = 0; y < 8; y++) {
                    const uint32_t pp = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_truncf(acc1[y]) + 128.0f, 1,
                                        __builtin_amdgcn_cvt_pk_u8_f32(__builtin_truncf(acc0[y]) + 128.0f, 0, 0u));
                    *reinterpret_cast<uint16_t *>(&s_pix[blk][y * 8 + x0]) = (uint16_t)pp;
                }
            } else { // extreme coefficients: the reference's int16 arithmetic, literally
#pragma unroll
                for (int y = 0; y < 8; y++) {
                    const int ta = (int)(int16_t)((int)(int16_t)(int)acc0[y] + 128);
                    const int tb = (int)(int16_t)((int)(int16_t)(int)acc1[y] + 128);
                    const uint32_t pa = (uint32_t)min(max(ta, 0), 255), pb = (uint32_t)min(max(tb, 0), 255);
                    *reinterpret_cast<uint16_t *>(&s_pix[blk][y * 8 + x0]) = (uint16_t)(pa | (pb << 8));
                }
            }
        }
    }
    __syncthreads();
#if defined(HJD_IDCT_CUT) && HJD_IDCT_CUT == 2 // stop after phase 2
    P.out[im.out_off + (size_t)lg * 256u + tid] = s_pix[tid % MAXGB][tid & 63];
    return;
#endif

    // phase 3: YCrCB_to_RGB24_Block8x8 (loadjpg.cpp:884-932) + output layout
    if (im.out_format == 1) color_store<true, CONF>(im, P.out + im.out_off, s_pix, tid, mrow, mcu_first, K, lg);
    else color_store<false, CONF>(im, P.out + im.out_off, s_pix, tid, mrow, mcu_first, K, lg);
}

// Test hook: the colour stage, alone, over every (Y, Cb, Cr) triple.  Output: 3 bytes
// R,G,B at index (Y << 16 | Cb << 8 | Cr) * 3, produced by the very functions k_idct_color uses.
__global__ __launch_bounds__(256) void k_color_table(uint8_t *out)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x; // (Y, Cb, quarter of Cr): four triples with consecutive Cr
    const uint32_t y = t >> 14, cb = (t >> 6) & 0xFF, cr0 = (t & 63u) * 4u;
    uint32_t d0, d1, d2;
    rgb12<false, false>(y * 0x01010101u, chroma_terms<false>(cb, cr0), chroma_terms<false>(cb, cr0 + 1), chroma_terms<false>(cb, cr0 + 2),
                        chroma_terms<false>(cb, cr0 + 3), d0, d1, d2);
    uint32_t *dst = reinterpret_cast<uint32_t *>(out + (size_t)((y << 16) | (cb << 8) | cr0) * 3u);
    dst[0] = d0;
    dst[1] = d1;
    dst[2] = d2;
}

int hjd_launch_idct_color(const HjdParams &P, uint32_t ngroups, hipStream_t s)
{
    if (!ngroups) return 0;
    if (P.conformant) hipLaunchKernelGGL(k_idct_color<true>, dim3(ngroups), dim3(256), 0, s, P);
    else hipLaunchKernelGGL(k_idct_color<false>, dim3(ngroups), dim3(256), 0, s, P);
    hipError_t e_ = hipGetLastError();
    return e_ == hipSuccess ? 0 : (int)e_;
}

int hjd_launch_color_table(uint8_t *d_out, hipStream_t s)
{
    hipLaunchKernelGGL(k_color_table, dim3((1u << 22) / 256), dim3(256), 0, s, d_out);
    hipError_t e_ = hipGetLastError();
    return e_ == hipSuccess ? 0 : (int)e_;
}
