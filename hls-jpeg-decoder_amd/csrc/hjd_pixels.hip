// hjd_pixels.hip — block back end (gfx950): dequantise, 8x8 IDCT, chroma upsample, YCbCr->RGB,
// output layout.
//   k_idct_color   DecodeSingleBlock + YCrCB_to_RGB24_Block8x8 (+ WriteBMP24 layout)
//                  (loadjpg.cpp:184-228, :867-932; openjpg.cpp:504-570)
//
// Compile with -ffp-contract=off: the reference's IDCT and colour arithmetic are fp32 with
// one rounding per operation; an FMA would change output bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hjd_device.h"
#include "hjd_kernels.h"
#include "hjd_kutil.h"

// ---- stage 4: dequantise + IDCT + upsample + colour + store -----------------------------------

// natural index -> zig-zag index (loadjpg.cpp:56-66)
__constant__ uint8_t c_zz[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42,
                                 3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
                                 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
                                 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

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

#define MAXGB (HJD_GROUP_MCUS * 6) // blocks per workgroup

__device__ __forceinline__ uint32_t clamp255(int v) { return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// ConvertYCrCbtoRGB as called at loadjpg.cpp:918 (:867-880), one rounding per operation
__device__ __forceinline__ void ycc_to_rgb(int y, int cb, int cr, uint32_t &r, uint32_t &g, uint32_t &b)
{
    const float fy = (float)y, fcb = (float)(cb - 128), fcr = (float)(cr - 128);
    const float red = fy + 1.402f * fcr;
    const float green = (fy - 0.34414f * fcb) - 0.71414f * fcr;
    const float blue = fy + 1.772f * fcb;
    r = clamp255((int)red);
    g = clamp255((int)green);
    b = clamp255((int)blue);
}

__global__ __launch_bounds__(256) void k_idct_color(HjdParams P)
{
    __shared__ int16_t s_coef[MAXGB][66]; // dequantised, index u*8+v (u = horizontal frequency); +2 pad: bank spread
    __shared__ uint64_t s_mask[MAXGB];    // bit u*8+v set where the coefficient is non-zero
    __shared__ __attribute__((aligned(8))) uint8_t s_pix[MAXGB][72]; // decoded 8x8 tiles, row-major; +8 pad: bank spread
    __shared__ __attribute__((aligned(16))) float s_cosT[64]; // [u][x] = COS[x][u]
    __shared__ __attribute__((aligned(16))) float s_cos[64];  // [v][y] = COS[y][v]  (same matrix, named for its use)
    __shared__ uint8_t s_qt[3][64];
    __shared__ uint8_t s_pos[64];
    __shared__ float s_kk[64];            // C(u)*C(v) by index u*8+v
    __shared__ uint8_t s_order[MAXGB];
    __shared__ uint32_t s_hist[66];
    __shared__ uint32_t s_boff[MAXGB + 1];
    __shared__ uint8_t s_comp[MAXGB];     // component (0 Y, 1 Cb, 2 Cr) of every block of the group

    const uint32_t tid = threadIdx.x;
    const uint32_t bid = blockIdx.x + P.off_group;
    const int ii = find_image(P.img, P.nimg, bid, [](const HjdImage &m) { return m.group0; });
    const HjdImage &im = P.img[ii];
    const uint32_t lg = bid - im.group0;
    const uint32_t mrow = lg / im.groups_per_row, gx = lg % im.groups_per_row;
    const uint32_t mcu_first = gx * HJD_GROUP_MCUS;
    const uint32_t K = min((uint32_t)HJD_GROUP_MCUS, im.mcux - mcu_first);
    const uint32_t nb = im.nb, nby = im.nby, hf = im.hf, vf = im.vf;
    const uint32_t KB = K * nb;
    const uint32_t b0 = (mrow * im.mcux + mcu_first) * nb;

    if (tid < 64) {
        s_cosT[tid] = c_cos[(tid & 7) * 8 + (tid >> 3)];
        s_cos[tid] = c_cos[(tid & 7) * 8 + (tid >> 3)];
    }
    if (tid < 192) s_qt[tid >> 6][tid & 63] = im.qt[tid >> 6][tid & 63];
    if (tid < KB) {
        const uint32_t mm = tid % nb;
        s_comp[tid] = (uint8_t)(mm < nby ? 0u : mm - nby + 1u);
    }
    if (tid < 64) s_pos[c_zz[tid]] = (uint8_t)((tid & 7) * 8 + (tid >> 3)); // zig-zag slot -> u*8+v (natural tid = v*8+u)
    if (tid < 64) { // C(u)*C(v): fl(C0*C0) = 0x1.fffffep-2 (not 0.5), C0 = 1/sqrtf(2) = 0x1.6a09e6p-1 (loadjpg.cpp:96-102,120)
        const uint32_t u = tid >> 3, v = tid & 7;
        s_kk[tid] = (u == 0) ? (v == 0 ? 0x1.fffffep-2f : 0x1.6a09e6p-1f) : (v == 0 ? 0x1.6a09e6p-1f : 1.0f);
    }
    __syncthreads();

    // phase 1: DequantizeBlock + DeZigZag + TransformArray (loadjpg.cpp:144-180) on the sparse form.
    // Non-zero dequantised values are scattered to their (u,v) slot and flagged in the block's LDS
    // mask.  int16 wrap of the product as the reference's (short)(int) cast.
    {
        const uint32_t *rec = P.rec + im.rec_off;
        const uint2 *binfo = P.blk + im.blk_off + b0; // (first record, predicted DC) per block
        uint2 bi = make_uint2(0u, 0u);
        if (tid <= KB) {
            bi = binfo[tid];
            s_boff[tid] = min(bi.x, im.rec_cap);
        }
        if (tid < KB) { // DC: predicted value from k_dcapply, quantiser slot 0
            const uint32_t comp = s_comp[tid];
            const int16_t nat = (int16_t)((int)(int16_t)bi.y * (int)s_qt[comp][0]);
            s_coef[tid][0] = nat;
            s_mask[tid] = nat != 0 ? 1ull : 0ull;
        }
        __syncthreads();
        // sixteen threads per block walk its record range [boff[b], boff[b+1]) (contiguous, so the team's
        // loads coalesce); r1 < r0 after a corrupt stream simply yields no work.  The first record of
        // every (thread, block) pair is requested before any is used, so that a workgroup pays the global
        // memory latency once instead of once per pass.
        constexpr int NPASS = MAXGB / 16;
        uint32_t w0[NPASS];
#pragma unroll
        for (int k = 0; k < NPASS; k++) {
            const uint32_t blk = (tid >> 4) + 16u * k;
            w0[k] = HJD_REC_PAD;
            if (blk < KB) {
                const uint32_t t = s_boff[blk] + (tid & 15u);
                if (t < s_boff[blk + 1]) w0[k] = rec[t];
            }
        }
#pragma unroll
        for (int k = 0; k < NPASS; k++) {
            const uint32_t blk = (tid >> 4) + 16u * k;
            if (blk >= KB) break;
            const uint32_t r1 = s_boff[blk + 1];
            const uint32_t comp = s_comp[blk];
            unsigned long long mk = 0;
            uint32_t w = w0[k];
            for (uint32_t t = s_boff[blk] + (tid & 15u); t < r1;) {
                if (w != HJD_REC_PAD) {
                    const uint32_t zz = (w >> 16) & 63u;
                    const int16_t nat = (int16_t)((int)(int16_t)(w & 0xFFFFu) * (int)s_qt[comp][zz]);
                    if (nat != 0) {
                        const uint32_t pos = s_pos[zz];
                        s_coef[blk][pos] = nat;
                        mk |= 1ull << pos;
                    }
                }
                t += 16;
                if (t < r1) w = rec[t]; // blocks with more than 16 records (rare)
            }
            if (mk) atomicOr(reinterpret_cast<unsigned long long *>(&s_mask[blk]), mk);
        }
    }
    __syncthreads();

    // Blocks are handed to the 4-lane teams in order of decreasing non-zero count, so that the sixteen
    // teams of a wave run loops of similar length: counting sort over the 65 possible counts.
    if (tid < 66) s_hist[tid] = 0;
    __syncthreads();
    uint32_t my_cnt = 0, my_slot = 0;
    if (tid < KB) {
        my_cnt = (uint32_t)__popcll((unsigned long long)s_mask[tid]);
        my_slot = atomicAdd(&s_hist[64u - my_cnt], 1u);
    }
    __syncthreads();
    if (tid < 64) { // exclusive prefix over buckets 0..64 (bucket 64 = empty blocks, placed last)
        const uint32_t h = s_hist[tid];
        const uint32_t inc = wave_incl_scan(h);
        s_hist[tid] = inc - h;
        if (tid == 63) s_hist[64] = inc;
    }
    __syncthreads();
    if (tid < KB) s_order[s_hist[64u - my_cnt] + my_slot] = (uint8_t)tid;
    __syncthreads();

    // phase 2: IDCT_calc / PerformIDCT (loadjpg.cpp:105-140).  Thread = (block, two output columns
    // x0, x0+1), 2 x 8 row accumulators.  Terms are added in the reference's order (u outer, v inner);
    // zero coefficients add +-0 and are skipped.  term = (((C(u)C(v)) * b) * COS[x][u]) * COS[y][v].
    {
        const uint32_t x0 = (tid & 3) * 2;
        for (uint32_t slot = tid >> 2; slot < KB; slot += 64) {
            const uint32_t blk = s_order[slot];
            float acc0[8], acc1[8];
#pragma unroll
            for (int y = 0; y < 8; y++) acc0[y] = acc1[y] = 0.0f;
            const uint64_t mk = s_mask[blk];
#pragma unroll
            for (int half = 0; half < 2; half++) { // u = 0..3, then u = 4..7: ascending set bits = the term order
                uint32_t m32 = half ? (uint32_t)(mk >> 32) : (uint32_t)mk;
                while (m32) {
                    const uint32_t j = half * 32u + (uint32_t)__ffs((int)m32) - 1u;
                    m32 &= m32 - 1u;
                    const uint32_t u = j >> 3, v = j & 7;
                    const float t1 = s_kk[j] * (float)s_coef[blk][j];
                    const float2 cx = *reinterpret_cast<const float2 *>(&s_cosT[u * 8 + x0]);
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
#pragma unroll
            for (int y = 0; y < 8; y++) {
                // (int)(0.25*sum) then short +128 then Clamp (loadjpg.cpp:123,136-137,219)
                const int16_t ta = (int16_t)(int)(acc0[y] * 0.25f), tb = (int16_t)(int)(acc1[y] * 0.25f);
                const uint32_t pa = clamp255((int16_t)(ta + 128)), pb = clamp255((int16_t)(tb + 128));
                *reinterpret_cast<uint16_t *>(&s_pix[blk][y * 8 + x0]) = (uint16_t)(pa | (pb << 8));
            }
        }
    }
    __syncthreads();

    // phase 3: YCrCB_to_RGB24_Block8x8 (loadjpg.cpp:884-932): nearest-neighbour chroma, clip at
    // the image edge, then either m_rgb layout or WriteBMP24 layout (openjpg.cpp:555-568)
    {
        const uint32_t W = im.width, H = im.height;
        const uint32_t mw = 8 * hf, mh = 8 * vf;
        // hf, vf are 1, 2 or 4 (hjd_parse_header): all index arithmetic below is shifts and masks.  The
        // unit grid is padded to HJD_GROUP_MCUS MCUs per row so that a unit's row and column are bit fields.
        const uint32_t lhf = hf >> 1, lvf = vf >> 1; // log2 for 1, 2, 4
        const uint32_t upr = K * 2 * hf;             // valid 4-pixel units per row
        const uint32_t lupr = 5 + lhf;               // log2(HJD_GROUP_MCUS * 2 * hf)
        const uint32_t nunits = mh << lupr;
        uint8_t *out = P.out + im.out_off;
        const bool bmp = im.out_format == 1;
        const uint32_t stride = im.out_stride;
        for (uint32_t t = tid; t < nunits; t += 256) {
            const uint32_t row = t >> lupr, ux = t & ((1u << lupr) - 1u);
            if (ux >= upr) continue;
            const uint32_t px = ux * 4;
            const uint32_t k = px >> (3 + lhf), xin = px & (mw - 1u);
            const uint32_t gy = mrow * mh + row;
            const uint32_t gxp = (mcu_first + k) * mw + xin;
            if (gy >= H || gxp >= W) continue; // loadjpg.cpp:907-908
            const uint32_t yblk = k * nb + ((row >> 3) << lhf) + (xin >> 3);
            const uint32_t ypix = *reinterpret_cast<const uint32_t *>(&s_pix[yblk][(row & 7) * 8 + (xin & 7)]);
            const uint32_t cy = row >> lvf;
            const uint8_t *pcb = &s_pix[k * nb + nby][cy * 8];
            const uint8_t *pcr = &s_pix[k * nb + nby + 1][cy * 8];
            uint32_t rr[4], gg[4], bb[4];
            if (hf == 2) {
                // 4:2:x — pixels 0,1 and 2,3 share a chroma sample (cx = x/2, loadjpg.cpp:912): the four
                // chroma products are formed once per sample; every operation is the same fp32 operation
                // ConvertYCrCbtoRGB performs ((float)(c - 128) == (float)c - 128.0f exactly)
                const uint32_t cb2 = *reinterpret_cast<const uint16_t *>(&pcb[xin >> 1]);
                const uint32_t cr2 = *reinterpret_cast<const uint16_t *>(&pcr[xin >> 1]);
#pragma unroll
                for (int h2 = 0; h2 < 2; h2++) {
                    const float fcb = (float)((cb2 >> (8 * h2)) & 0xFF) - 128.0f;
                    const float fcr = (float)((cr2 >> (8 * h2)) & 0xFF) - 128.0f;
                    const float pa = 1.402f * fcr, pb = 0.34414f * fcb, pc = 0.71414f * fcr, pd = 1.772f * fcb;
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int j = 2 * h2 + q;
                        const float fy = (float)((ypix >> (8 * j)) & 0xFF);
                        rr[j] = clamp255((int)(fy + pa));
                        gg[j] = clamp255((int)((fy - pb) - pc));
                        bb[j] = clamp255((int)(fy + pd));
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t cx = (xin + j) >> lhf; // loadjpg.cpp:912
                    ycc_to_rgb((int)((ypix >> (8 * j)) & 0xFF), (int)pcb[cx], (int)pcr[cx], rr[j], gg[j], bb[j]);
                }
            }
            const uint32_t nvalid = min(4u, W - gxp);
            uint8_t *dst;
            if (bmp) dst = out + 54 + (size_t)(H - 1 - gy) * stride + (size_t)gxp * 3;
            else dst = out + (size_t)gy * stride + (size_t)gxp * 3;
            uint32_t c0[4], c2[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                c0[j] = bmp ? bb[j] : rr[j];
                c2[j] = bmp ? rr[j] : bb[j];
            }
            if (nvalid == 4 && ((reinterpret_cast<uintptr_t>(dst) & 3) == 0)) {
                struct __attribute__((packed, aligned(4))) Px12 { uint32_t a, b, c; };
                Px12 v12;
                v12.a = c0[0] | (gg[0] << 8) | (c2[0] << 16) | (c0[1] << 24);
                v12.b = gg[1] | (c2[1] << 8) | (c0[2] << 16) | (gg[2] << 24);
                v12.c = c2[2] | (c0[3] << 8) | (gg[3] << 16) | (c2[3] << 24);
                *reinterpret_cast<Px12 *>(dst) = v12; // one 12-byte store: a wave writes 768 contiguous bytes
            } else {
                for (uint32_t j = 0; j < nvalid; j++) {
                    dst[j * 3 + 0] = (uint8_t)c0[j];
                    dst[j * 3 + 1] = (uint8_t)gg[j];
                    dst[j * 3 + 2] = (uint8_t)c2[j];
                }
            }
            if (bmp && gxp + nvalid == W) { // row padding (openjpg.cpp:563-567), written as zeros
                for (uint32_t q = 3 * W; q < stride; q++) out[54 + (size_t)(H - 1 - gy) * stride + q] = 0;
            }
        }
        if (bmp && lg == 0 && tid < 54) {
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
}

int hjd_launch_idct_color(const HjdParams &P, uint32_t ngroups, hipStream_t s)
{
    if (!ngroups) return 0;
    hipLaunchKernelGGL(k_idct_color, dim3(ngroups), dim3(256), 0, s, P);
    hipError_t e_ = hipGetLastError();
    return e_ == hipSuccess ? 0 : (int)e_;
}
