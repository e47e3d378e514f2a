// mfma_hazard.hip -- which register reuse around v_mfma_f32_16x16x4_f32 is safe on gfx950?  (round 6: hipcc-generated code for the first
// layer's float32 MFMA kernel produced run-to-run different results in one accumulator chain.)  Hand-placed registers, inline asm:
//   T0  reference: in-place accumulate, long wait, read.
//   T1  vDst partially overlaps SrcC, dst BELOW (v[32:35] <- C v[34:37]) -- what hipcc emits when it slides a chain into dying registers
//   T2  vDst partially overlaps SrcC, dst ABOVE (v[36:39] <- C v[34:37])
//   T3  vDst contains SrcB (v[40:43] <- A v8, B v41, C v[50:53])
//   T4  vDst contains SrcA
//   T5<N>  WAR on SrcC behind a QUEUE: four independent MFMAs, N x s_nop 0, then v_mov into the LAST one's SrcC registers
//   T6<N>  WAR on SrcB behind a queue: four independent MFMAs (own B registers), N x s_nop 0, then v_mov into the last one's SrcB
//   T7<N>  RAW: four independent MFMAs, N x s_nop 0, then v_mov FROM the last one's vDst (is the read interlocked?)
// Build: hipcc --offload-arch=gfx950 -O3 -o ab/mfma_hazard scripts/micro/mfma_hazard.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define LOAD_ABC                                                                                                  \
    "global_load_dword v8, %1, off\n global_load_dword v9, %2, off\n global_load_dwordx4 v[34:37], %3, off\n s_waitcnt vmcnt(0)\n"
#define WAIT_ALL "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
#define CLOB "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "memory"

template <int T, int N>
__global__ void __launch_bounds__(1024) k(const float *a, const float *b, const float *c, float *out)
{
    const int lane = threadIdx.x & 63;
    const float *pa = a + lane, *pb = b + lane, *pc = c + lane * 4;
    float *po = out + ((blockIdx.x == 0 ? threadIdx.x >> 6 : 16) * 64 + lane) * 4;   // block 0's waves are checked
    if constexpr (T == 0)
        asm volatile(LOAD_ABC "v_mfma_f32_16x16x4_f32 v[34:37], v8, v9, v[34:37]\n" WAIT_ALL "global_store_dwordx4 %0, v[34:37], off\n s_waitcnt vmcnt(0)\n"
                     :: "v"(po), "v"(pa), "v"(pb), "v"(pc) : CLOB);
    if constexpr (T == 1)
        asm volatile(LOAD_ABC "v_mfma_f32_16x16x4_f32 v[32:35], v8, v9, v[34:37]\n" WAIT_ALL "global_store_dwordx4 %0, v[32:35], off\n s_waitcnt vmcnt(0)\n"
                     :: "v"(po), "v"(pa), "v"(pb), "v"(pc) : CLOB);
    if constexpr (T == 2)
        asm volatile(LOAD_ABC "v_mfma_f32_16x16x4_f32 v[36:39], v8, v9, v[34:37]\n" WAIT_ALL "global_store_dwordx4 %0, v[36:39], off\n s_waitcnt vmcnt(0)\n"
                     :: "v"(po), "v"(pa), "v"(pb), "v"(pc) : CLOB);
    if constexpr (T == 3)
        asm volatile(LOAD_ABC "v_mov_b32 v41, v9\n" WAIT_ALL "v_mfma_f32_16x16x4_f32 v[40:43], v8, v41, v[34:37]\n" WAIT_ALL
                     "global_store_dwordx4 %0, v[40:43], off\n s_waitcnt vmcnt(0)\n" :: "v"(po), "v"(pa), "v"(pb), "v"(pc) : CLOB);
    if constexpr (T == 4)
        asm volatile(LOAD_ABC "v_mov_b32 v41, v8\n" WAIT_ALL "v_mfma_f32_16x16x4_f32 v[40:43], v41, v9, v[34:37]\n" WAIT_ALL
                     "global_store_dwordx4 %0, v[40:43], off\n s_waitcnt vmcnt(0)\n" :: "v"(po), "v"(pa), "v"(pb), "v"(pc) : CLOB);
    if constexpr (T == 5 || T == 6 || T == 7) {
        asm volatile(LOAD_ABC
                     "v_mov_b32 v10, v34\n v_mov_b32 v11, v35\n v_mov_b32 v12, v36\n v_mov_b32 v13, v37\n"
                     "v_mov_b32 v14, v34\n v_mov_b32 v15, v35\n v_mov_b32 v16, v36\n v_mov_b32 v17, v37\n"
                     "v_mov_b32 v18, v34\n v_mov_b32 v19, v35\n v_mov_b32 v20, v36\n v_mov_b32 v21, v37\n"
                     "v_mov_b32 v22, v34\n v_mov_b32 v23, v35\n v_mov_b32 v24, v36\n v_mov_b32 v25, v37\n"
                     "v_mov_b32 v26, v9\n v_mov_b32 v27, v9\n v_mov_b32 v28, v9\n v_mov_b32 v29, v9\n"
                     "v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n" WAIT_ALL
                     "v_mfma_f32_16x16x4_f32 v[30:33], v8, v26, v[10:13]\n"
                     "v_mfma_f32_16x16x4_f32 v[38:41], v8, v27, v[14:17]\n"
                     "v_mfma_f32_16x16x4_f32 v[42:45], v8, v28, v[18:21]\n"
                     "v_mfma_f32_16x16x4_f32 v[46:49], v8, v29, v[22:25]\n"
                     :: "v"(po), "v"(pa), "v"(pb), "v"(pc) : CLOB);
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("s_nop 0" ::: CLOB);
        if constexpr (T == 5)
            asm volatile("v_mov_b32 v22, 0x7fc00000\n v_mov_b32 v23, 0x7fc00000\n v_mov_b32 v24, 0x7fc00000\n v_mov_b32 v25, 0x7fc00000\n" WAIT_ALL
                         "global_store_dwordx4 %0, v[46:49], off\n s_waitcnt vmcnt(0)\n" :: "v"(po) : CLOB);
        if constexpr (T == 6)
            asm volatile("v_mov_b32 v29, 0x7fc00000\n" WAIT_ALL "global_store_dwordx4 %0, v[46:49], off\n s_waitcnt vmcnt(0)\n" :: "v"(po) : CLOB);
        if constexpr (T == 7)
            asm volatile("v_mov_b32 v50, v46\n v_mov_b32 v51, v47\n v_mov_b32 v52, v48\n v_mov_b32 v53, v49\n" WAIT_ALL
                         "global_store_dwordx4 %0, v[50:53], off\n s_waitcnt vmcnt(0)\n" :: "v"(po) : CLOB);
    }
    if constexpr (T == 8 || T == 9 || T == 10) {
        asm volatile(LOAD_ABC WAIT_ALL
                     "v_mfma_f32_16x16x4_f32 v[60:63], v8, v9, v[34:37]\n"
                     "v_mfma_f32_16x16x4_f32 v[64:67], v8, v9, v[34:37]\n"
                     "v_mfma_f32_16x16x4_f32 v[68:71], v8, v9, v[34:37]\n"
                     "v_mfma_f32_16x16x4_f32 v[72:75], v8, v9, v[34:37]\n"
                     "v_mfma_f32_16x16x4_f32 v[76:79], v8, v9, v[34:37]\n"
                     "v_mfma_f32_16x16x4_f32 v[80:83], v8, v9, v[34:37]\n"
                     "v_mfma_f32_16x16x4_f32 v[84:87], v8, v9, v[34:37]\n"
                     "v_mfma_f32_16x16x4_f32 v[88:91], v8, v9, v[34:37]\n"
                     "v_mfma_f32_16x16x4_f32 v[92:95], v8, v9, v[34:37]\n"
                     "v_mfma_f32_16x16x4_f32 v[96:99], v8, v9, v[34:37]\n"
                     "v_mfma_f32_16x16x4_f32 v[100:103], v8, v9, v[34:37]\n"
                     "v_mfma_f32_16x16x4_f32 v[104:107], v8, v9, v[34:37]\n"
                     "v_mfma_f32_16x16x4_f32 v[108:111], v8, v9, v[34:37]\n"
                     "v_mfma_f32_16x16x4_f32 v[112:115], v8, v9, v[34:37]\n"
                     "v_mfma_f32_16x16x4_f32 v[116:119], v8, v9, v[34:37]\n"
                     "v_mfma_f32_16x16x4_f32 v[120:123], v8, v9, v[34:37]\n"
                     :: "v"(po), "v"(pa), "v"(pb), "v"(pc) : CLOB);
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("s_nop 0" ::: CLOB);
        if constexpr (T == 8)
            asm volatile("v_mov_b32 v9, 0x7fc00000\n" WAIT_ALL WAIT_ALL WAIT_ALL WAIT_ALL "global_store_dwordx4 %0, v[120:123], off\n s_waitcnt vmcnt(0)\n" :: "v"(po) : CLOB);
        if constexpr (T == 9)
            asm volatile("v_mov_b32 v8, 0x7fc00000\n" WAIT_ALL WAIT_ALL WAIT_ALL WAIT_ALL "global_store_dwordx4 %0, v[120:123], off\n s_waitcnt vmcnt(0)\n" :: "v"(po) : CLOB);
        if constexpr (T == 10)
            asm volatile("v_mov_b32 v34, 0x7fc00000\n v_mov_b32 v35, 0x7fc00000\n v_mov_b32 v36, 0x7fc00000\n v_mov_b32 v37, 0x7fc00000\n" WAIT_ALL WAIT_ALL WAIT_ALL WAIT_ALL
                         "global_store_dwordx4 %0, v[120:123], off\n s_waitcnt vmcnt(0)\n" :: "v"(po) : CLOB);
    }
}

static std::vector<float> ha(64), hb(64), hc(256), want(256);
static float *da, *db, *dc, *dout;
static int WAVES = 1;

template <int T, int N>
static void run(const char *what)
{
    (void)hipMemset(dout, 0xff, 17 * 1024);
    int bad_total = 0, runs = 0;
    std::vector<float> got(16 * 256);
    unsigned lanes_bad = 0, elems_bad = 0;
    for (int rep = 0; rep < 200; ++rep) {
        k<T, N><<<256, WAVES * 64>>>(da, db, dc, dout);        // (256 workgroups: block 0's waves are checked, the rest make the chip busy)
        (void)hipMemcpy(got.data(), dout, WAVES * 1024, hipMemcpyDeviceToHost);
        ++runs;
        for (int wv = 0; wv < WAVES; ++wv)
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r)
                if (!(got[wv * 256 + l * 4 + r] == want[l * 4 + r])) {
                    ++bad_total;
                    lanes_bad |= 1u << (l & 15);
                    elems_bad |= 1u << r;
                }
    }
    printf("waves/block %2d T%d N=%2d %-58s: %s  (%d wrong values in %d runs; lanes&15 mask %04x, element mask %x)\n", WAVES, T, N, what, bad_total ? "WRONG" : "ok", bad_total,
           runs, lanes_bad, elems_bad);
}

int main()
{
    srand(3);
    for (auto &v : ha) v = (float)(rand() % 7 - 3);
    for (auto &v : hb) v = (float)(rand() % 7 - 3);
    for (auto &v : hc) v = (float)(rand() % 17 - 8);
    // D[i][j] = sum_k A[i][k] B[k][j] + C[i][j];  A: lane (i = l % 16, k = l / 16);  B: lane (k = l / 16, j = l % 16);  C/D: lane (j = l % 16, i = 4 (l / 16) + r)
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int j = l % 16, i = 4 * (l / 16) + r;
            float s = hc[l * 4 + r];
            for (int kk = 0; kk < 4; ++kk) s += ha[kk * 16 + i] * hb[kk * 16 + j];
            want[l * 4 + r] = s;
        }
    (void)hipMalloc(&da, 256); (void)hipMalloc(&db, 256); (void)hipMalloc(&dc, 1024); (void)hipMalloc(&dout, 17 * 1024);
    (void)hipMemcpy(da, ha.data(), 256, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, hb.data(), 256, hipMemcpyHostToDevice);
    (void)hipMemcpy(dc, hc.data(), 1024, hipMemcpyHostToDevice);
    for (int wvs : {1, 16}) {
    WAVES = wvs;
    run<0, 0>("in place");
    run<1, 0>("vDst v[32:35] <- SrcC v[34:37] (partial overlap, below)");
    run<2, 0>("vDst v[36:39] <- SrcC v[34:37] (partial overlap, above)");
    run<3, 0>("vDst v[40:43] contains SrcB v41");
    run<4, 0>("vDst v[40:43] contains SrcA v41");
    run<5, 0>("4 queued MFMAs, v_mov into the last one's SrcC");
    run<5, 2>("4 queued MFMAs, v_mov into the last one's SrcC");
    run<5, 4>("4 queued MFMAs, v_mov into the last one's SrcC");
    run<5, 8>("4 queued MFMAs, v_mov into the last one's SrcC");
    run<5, 16>("4 queued MFMAs, v_mov into the last one's SrcC");
    run<5, 24>("4 queued MFMAs, v_mov into the last one's SrcC");
    run<5, 32>("4 queued MFMAs, v_mov into the last one's SrcC");
    run<6, 0>("4 queued MFMAs, v_mov into the last one's SrcB");
    run<6, 4>("4 queued MFMAs, v_mov into the last one's SrcB");
    run<6, 16>("4 queued MFMAs, v_mov into the last one's SrcB");
    run<6, 32>("4 queued MFMAs, v_mov into the last one's SrcB");
    run<7, 0>("4 queued MFMAs, v_mov FROM the last one's vDst (RAW)");
    run<7, 4>("4 queued MFMAs, v_mov FROM the last one's vDst (RAW)");
    run<7, 16>("4 queued MFMAs, v_mov FROM the last one's vDst (RAW)");
    run<7, 32>("4 queued MFMAs, v_mov FROM the last one's vDst (RAW)");
    run<8, 0>("16 queued MFMAs sharing A/B/C, v_mov into SrcB");
    run<8, 4>("16 queued MFMAs sharing A/B/C, v_mov into SrcB");
    run<8, 16>("16 queued MFMAs sharing A/B/C, v_mov into SrcB");
    run<8, 48>("16 queued MFMAs sharing A/B/C, v_mov into SrcB");
    run<9, 0>("16 queued MFMAs sharing A/B/C, v_mov into SrcA");
    run<9, 4>("16 queued MFMAs sharing A/B/C, v_mov into SrcA");
    run<9, 16>("16 queued MFMAs sharing A/B/C, v_mov into SrcA");
    run<9, 48>("16 queued MFMAs sharing A/B/C, v_mov into SrcA");
    run<10, 0>("16 queued MFMAs sharing A/B/C, v_mov into SrcC");
    run<10, 4>("16 queued MFMAs sharing A/B/C, v_mov into SrcC");
    run<10, 16>("16 queued MFMAs sharing A/B/C, v_mov into SrcC");
    run<10, 48>("16 queued MFMAs sharing A/B/C, v_mov into SrcC");
    }
    return 0;
}
