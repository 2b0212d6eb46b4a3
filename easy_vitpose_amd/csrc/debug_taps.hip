// vp_dbg_*: parity taps (ONE kernel on host fp32 data, used by tests/) and, in the measurement build (-DVP_TOOLS), the timing taps tools/ load.
#include "api_internal.h"

using namespace vpi;

// ---------------------------------------------------------------------------
// Debug / parity taps: run ONE kernel on host fp32 data (operands are rounded to
// `dtype` exactly as the production packer / producers do).  Used by tests/ only.
// ---------------------------------------------------------------------------
namespace {
vp_ctx* dbg_ctx(int device, int dtype) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        g_create_error = "no HIP device available (no CPU fallback)";
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    vp_ctx* c = new vp_ctx();
    c->cfg.device_id = device;
    c->dtype = dtype == VP_DTYPE_F16 ? vp::DT_F16 : vp::DT_BF16;
    apply_gemm_tuning(c);
    return c;
}
int dbg_finish(vp_ctx* c, int rc) {
    if (rc) g_create_error = c->err;
    vp_destroy(c);
    return rc;
}
// device 16-bit -> host fp32
int download16(vp_ctx* c, const uint16_t* d, float* out, size_t n) {
    std::vector<uint16_t> t(n);
    HIPCHK(c, hipMemcpy(t.data(), d, n * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) {
        if (c->dtype == vp::DT_BF16) {
            uint32_t u = (uint32_t)t[i] << 16;
            std::memcpy(&out[i], &u, 4);
        } else {
            const uint32_t h = t[i], sign = (h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff;
            uint32_t u;
            if (e == 0) {
                if (m == 0) u = sign;
                else { int sh = 0; uint32_t mm = m; while (!(mm & 0x400)) { mm <<= 1; ++sh; }
                       u = sign | ((uint32_t)(113 - sh) << 23) | ((mm & 0x3ff) << 13); }
            } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
            else u = sign | ((e + 112) << 23) | (m << 13);
            std::memcpy(&out[i], &u, 4);
        }
    }
    return VP_OK;
}
}  // namespace

extern "C" {

// out = epilogue(A[M,K] . W[N,K]^T): epi 0 bias->16bit, 1 bias+gelu->16bit, 2 bias+aux[M,N]->fp32, 3 aux[m%192]->fp32
VP_API int vp_dbg_gemm(int32_t device, int32_t dtype, int32_t epi, int32_t M, int32_t N, int32_t K, const float* A,
                       const float* W, const float* bias, const float* aux, float* out) {
    if (epi < 0 || epi > 3 || M <= 0 || N <= 0 || K <= 0 || K % 64) return fail(nullptr, VP_ERR_INVALID, "bad gemm test shape");
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    uint16_t *dA, *dW, *dO16 = nullptr;
    float *dB, *dAux = nullptr, *dO32 = nullptr;
    int rc;
    const size_t MN = (size_t)M * N;
    if ((rc = upload_mat(c, &dA, A, M, K, M))) return dbg_finish(c, rc);
    if ((rc = upload_mat(c, &dW, W, N, K, pad128(N)))) return dbg_finish(c, rc);
    if ((rc = upload_f32(c, &dB, bias, N, pad128(N)))) return dbg_finish(c, rc);
    if (epi >= 2) {
        if ((rc = upload_f32(c, &dAux, aux, epi == 2 ? MN : (size_t)192 * N))) return dbg_finish(c, rc);
        if ((rc = dalloc(c, &dO32, MN))) return dbg_finish(c, rc);
    } else if ((rc = dalloc(c, &dO16, MN))) return dbg_finish(c, rc);
    if ((rc = dalloc(c, &c->zero, (size_t)256))) return dbg_finish(c, rc);
    rc = gemm(c, 0, epi, dA, dW, dB, epi >= 2 ? (void*)dO32 : (void*)dO16, dAux, M, N, K, N);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(c, VP_ERR_HIP, "gemm kernel failed");
    if (!rc) {
        if (epi >= 2) { if (hipMemcpy(out, dO32, MN * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(c, VP_ERR_HIP, "D2H"); }
        else rc = download16(c, dO16, out, MN);
    }
    return dbg_finish(c, rc);
}

// qkv [B*192, 3*D] fp32 -> out [B*192, D] fp32 (attention core, vit.py:167-176)
VP_API int vp_dbg_attention(int32_t device, int32_t dtype, int32_t B, int32_t D, int32_t heads, const float* qkv, float* out) {
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    uint16_t *dq, *dout;
    int rc;
    const size_t M = (size_t)B * 192;
    if ((rc = upload_mat(c, &dq, qkv, M, 3 * (size_t)D, M))) return dbg_finish(c, rc);
    if ((rc = dalloc(c, &dout, M * D))) return dbg_finish(c, rc);
    hipError_t e = vp::attention_launch(c->dtype, dq, dout, B, D, heads, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("attention: ") + hipGetErrorString(e)));
    return dbg_finish(c, download16(c, dout, out, M * D));
}

// attn.qkv + attention core in one kernel (qkvattn.hip): x [2 npairs 192, D] (rounded to dtype), Wqkv [3D, D], bias [3D] -> out [M, D] (as fp32).
// Run with neutral LayerNorm statistics (mean 0, rstd 1, row sums 0: ln_fold(acc, 0, 0, 1, b) == acc + b exactly), so the result must equal
// vp_dbg_gemm(epi 0) followed by vp_dbg_attention bit for bit.
VP_API int vp_dbg_qkvattn(int32_t device, int32_t dtype, int32_t npairs, int32_t D, int32_t heads, const float* x, const float* W, const float* bias, float* out) {
    if (npairs <= 0 || D <= 0 || heads <= 0 || !x || !W || !bias || !out) return fail(nullptr, VP_ERR_INVALID, "bad qkvattn test shape");
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    const size_t M = (size_t)npairs * 384;
    uint16_t *dx, *dw, *dwh, *dy;
    float *db, *dbh, *ds, *dsh, *drow;
    int rc;
    std::vector<float> zeros(3 * (size_t)D, 0.f), row(2 * M);
    for (size_t m = 0; m < M; ++m) { row[2 * m] = 0.f; row[2 * m + 1] = 1.f; }
    if ((rc = upload_mat(c, &dx, x, M, D, M)) || (rc = upload_mat(c, &dw, W, 3 * (size_t)D, D, pad128(3 * (size_t)D))) || (rc = upload_f32(c, &db, bias, 3 * (size_t)D)) ||
        (rc = upload_f32(c, &ds, zeros.data(), 3 * (size_t)D)) || (rc = upload_f32(c, &drow, row.data(), 2 * M)) || (rc = dalloc(c, &dwh, 3 * (size_t)D * D)) ||
        (rc = dalloc(c, &dbh, 3 * (size_t)D)) || (rc = dalloc(c, &dsh, 3 * (size_t)D)) || (rc = dalloc(c, &dy, M * D)))
        return dbg_finish(c, rc);
    if (heads * 80 == D) {   // head dim 80: gemm8.hip EPI_QKV_ATTN on the 192 x 256 tile (one crop x one head), head-major weights of heads * 256 rows
        uint16_t* dwh80; float *dbh80, *dsh80;
        const size_t rows = (size_t)heads * 256;
        if ((rc = dalloc(c, &dwh80, rows * D)) || (rc = dalloc(c, &dbh80, rows)) || (rc = dalloc(c, &dsh80, rows))) return dbg_finish(c, rc);
        hipError_t e8 = vp::qkv_head_major80_launch(dw, db, ds, dwh80, dbh80, dsh80, D, D, heads, nullptr);
        vp::GemmArgs g80{};
        g80.A = dx; g80.W = dwh80; g80.bias = dbh80; g80.ln_s = dsh80; g80.rowstat = drow; g80.out = dy;
        g80.M = (int)M; g80.N = heads * 256; g80.K = D; g80.ldo = D; g80.w_rows = heads * 256; g80.variant = 18;
        g80.attn_scale_log2e = (1.0f / sqrtf(80.0f)) * 1.4426950408889634f;
        if (e8 == hipSuccess && !vp::gemm8_supported(vp::EPI_QKV_ATTN, g80, 256, 192)) return dbg_finish(c, fail(c, VP_ERR_INVALID, "shape not supported by the fused qkv + attention tile (head dim 80)"));
        if (e8 == hipSuccess) e8 = vp::gemm_launch(c->dtype, vp::EPI_QKV_ATTN, g80, nullptr);
        if (e8 == hipSuccess) e8 = hipDeviceSynchronize();
        if (e8 != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("qkvattn (head dim 80): ") + hipGetErrorString(e8)));
        return dbg_finish(c, download16(c, dy, out, M * D));
    }
    hipError_t e = vp::qkv_head_major_launch(dw, db, ds, dwh, dbh, dsh, D, D, nullptr);
    vp::QkvAttnArgs qa{};
    qa.x_hi = dx; qa.wh = dwh; qa.bh = dbh; qa.sh = dsh; qa.rowstat = drow; qa.y = dy; qa.npairs = npairs; qa.ncrops = 2 * npairs; qa.heads = heads; qa.D = D;
    const float scale = 1.0f / sqrtf(64.0f);
    qa.scale_log2e = scale * 1.4426950408889634f;
    if (e == hipSuccess && !vp::qkvattn_supported(qa)) return dbg_finish(c, fail(c, VP_ERR_INVALID, "shape not supported by the fused qkv + attention kernel"));
    if (e == hipSuccess) e = vp::qkvattn_launch(c->dtype, qa, nullptr, nullptr, 0);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("qkvattn: ") + hipGetErrorString(e)));
    return dbg_finish(c, download16(c, dy, out, M * D));
}

#ifdef VP_TOOLS
// tools/qkvattn_phases.py: average milliseconds of the fused qkv + attention kernel on random operands, optionally with phases compiled out
VP_API int vp_dbg_qkvattn_bench(int32_t device, int32_t npairs, int32_t D, int32_t heads, int32_t iters, int32_t ablate, float* ms_out) {
    vp_ctx* c = dbg_ctx(device, VP_DTYPE_F16);
    if (!c) return VP_ERR_HIP;
    const size_t M = (size_t)npairs * 384;
    uint16_t *dx, *dwh, *dy;
    float *dbh, *dsh, *drow;
    int rc;
    if ((rc = dalloc(c, &dx, M * D)) || (rc = dalloc(c, &dwh, 3 * (size_t)D * D)) || (rc = dalloc(c, &dy, M * D)) || (rc = dalloc(c, &dbh, 3 * (size_t)D)) ||
        (rc = dalloc(c, &dsh, 3 * (size_t)D)) || (rc = dalloc(c, &drow, 2 * M)))
        return dbg_finish(c, rc);
    vp::fill_random16(c->dtype, dx, M * D, 1u, nullptr);
    vp::fill_random16(c->dtype, dwh, 3 * (size_t)D * D, 2u, nullptr);
    hipMemset(dbh, 0, 3 * (size_t)D * 4); hipMemset(dsh, 0, 3 * (size_t)D * 4); hipMemset(drow, 0, 2 * M * 4);
    vp::QkvAttnArgs qa{};
    qa.x_hi = dx; qa.wh = dwh; qa.bh = dbh; qa.sh = dsh; qa.rowstat = drow; qa.y = dy; qa.npairs = npairs; qa.ncrops = 2 * npairs; qa.heads = heads; qa.D = D; qa.ablate = ablate;
    qa.scale_log2e = 0.125f * 1.4426950408889634f;
    vp::GemmArgs g80{};   // head dim 80: gemm8.hip EPI_QKV_ATTN (heads * 256 head-major rows: the 3 D^2 buffer is larger than heads * 256 * D)
    const bool h80 = heads * 80 == D;
    g80.A = dx; g80.W = dwh; g80.bias = dbh; g80.ln_s = dsh; g80.rowstat = drow; g80.out = dy;
    g80.M = (int)M; g80.N = heads * 256; g80.K = D; g80.ldo = D; g80.w_rows = heads * 256; g80.variant = 18; g80.ablate = ablate;
    g80.attn_scale_log2e = (1.0f / sqrtf(80.0f)) * 1.4426950408889634f;
    auto launch = [&]() { return h80 ? vp::gemm_launch(c->dtype, vp::EPI_QKV_ATTN, g80, nullptr) : vp::qkvattn_launch(c->dtype, qa, nullptr, nullptr, 0); };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters && e == hipSuccess; ++i) e = launch();
    hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("qkvattn bench: ") + hipGetErrorString(e)));
    return dbg_finish(c, VP_OK);
}
#endif

// LayerNorm(eps 1e-6): x [M,D] fp32 -> out16 (as fp32) [M,D] and out32 [M,D]
VP_API int vp_dbg_layernorm(int32_t device, int32_t dtype, int32_t M, int32_t D, const float* x, const float* gamma,
                            const float* beta, float* out16, float* out32) {
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    float *dx, *dg, *db, *d32;
    uint16_t* d16;
    int rc;
    const size_t MD = (size_t)M * D;
    if ((rc = upload_f32(c, &dx, x, MD)) || (rc = upload_f32(c, &dg, gamma, D)) || (rc = upload_f32(c, &db, beta, D)) ||
        (rc = dalloc(c, &d32, MD)) || (rc = dalloc(c, &d16, MD))) return dbg_finish(c, rc);
    hipError_t e = vp::layernorm_launch(c->dtype, dx, dg, db, d16, d32, M, D, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out32, d32, MD * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("layernorm: ") + hipGetErrorString(e)));
    return dbg_finish(c, download16(c, d16, out16, MD));
}

// ConvTranspose2d(Cin,256,4,2,1,bias=False)+BN(eval)+ReLU on NHWC x [B,Hin,Win,Cin] fp32 -> NHWC [B,2Hin,2Win,256] fp32.
// tensors = {"keypoint_head.deconv_layers.0.weight", ".1.weight", ".1.bias", ".1.running_mean", ".1.running_var"}
VP_API int vp_dbg_deconv(int32_t device, int32_t dtype, int32_t B, int32_t Hin, int32_t Win, int32_t Cin, const float* x,
                         const vp_tensor_desc* tensors, int32_t n_tensors, float* out) {
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    Lookup lk;
    lk.c = c;
    for (int i = 0; i < n_tensors; ++i) lk.map[tensors[i].name] = &tensors[i];
    uint16_t *dx, *dw, *dout;
    float* db;
    int rc;
    const size_t Min = (size_t)B * Hin * Win;
    if ((rc = pack_deconv(c, lk, 0, Cin, &dw, &db))) return dbg_finish(c, rc);
    if ((rc = upload_mat(c, &dx, x, Min, Cin, Min))) return dbg_finish(c, rc);
    if ((rc = dalloc(c, &dout, Min * 4 * 256))) return dbg_finish(c, rc);
    if ((rc = dalloc(c, &c->zero, (size_t)256))) return dbg_finish(c, rc);
    if (hipMemset(c->zero, 0, 512) != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, "memset"));
    rc = gemm(c, 0, vp::EPI_DECONV, dx, dw, db, dout, nullptr, (int)Min, 256, 4 * Cin, 256, Hin, Win, Cin);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(c, VP_ERR_HIP, "deconv kernel failed");
    if (!rc) rc = download16(c, dout, out, Min * 4 * 256);
    return dbg_finish(c, rc);
}


#ifdef VP_TOOLS
// tools/gemm_timeline.py: one persistent launch of the qkv / fc1 shape with per-tile phase stamps (shader cycles) of wave 0 of
// every workgroup: stamps[wg][tile][8] = (main loop start, main loop end, epilogue end, 5 stamps inside k-step 5: top, after the
// barrier, after the global_load_lds issues, after the first MFMA block, end), up to 32 tiles per workgroup.
VP_API int vp_dbg_gemm_timeline(int32_t device, int32_t dtype, int32_t epi, int32_t M, int32_t N, int32_t K, uint64_t* stamps,
                                int32_t max_wg) {
    if ((epi != 0 && epi != 1) || !stamps) return fail(nullptr, VP_ERR_INVALID, "bad timeline request");
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    uint16_t *dA, *dW, *dO;
    float* dB;
    unsigned long long* dS;
    int rc;
    const size_t wrows = pad128(N), nst = (size_t)max_wg * 32 * 8;
    if ((rc = dalloc(c, &dA, (size_t)M * K)) || (rc = dalloc(c, &dW, wrows * K)) || (rc = dalloc(c, &dB, wrows)) ||
        (rc = dalloc(c, &dO, (size_t)M * N)) || (rc = dalloc(c, &dS, nst)) || (rc = dalloc(c, &c->zero, (size_t)256)))
        return dbg_finish(c, rc);
    vp::fill_random16(c->dtype, dA, (size_t)M * K, 1u, nullptr);
    vp::fill_random16(c->dtype, dW, wrows * K, 2u, nullptr);
    hipMemset(dB, 0, wrows * 4);
    hipMemset(dS, 0, nst * 8);
    vp::GemmArgs g{};
    g.A = dA; g.W = dW; g.bias = dB; g.out = dO; g.M = M; g.N = N; g.K = K; g.ldo = N; g.zero = c->zero;
    g.w_rows = (int)wrows; g.variant = 8; g.group_m = 8; g.persist = 1;
    hipError_t e = vp::gemm_launch(c->dtype, epi, g, nullptr);          // warm
    g.ablate = 32 | (getenv("VP_TL_ABL") ? atoi(getenv("VP_TL_ABL")) : 0); g.stats_out = (float*)dS;
    if (e == hipSuccess) e = vp::gemm_launch(c->dtype, epi, g, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(stamps, dS, nst * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("timeline: ") + hipGetErrorString(e)));
    return dbg_finish(c, VP_OK);
}
#endif  // VP_TOOLS

#ifdef VP_TOOLS   // timing tap of the measurement build (include/vitpose_hip_tools.h)
// Time `iters` launches of one GEMM configuration on random device operands (HIP events).
// epi as in vp_dbg_gemm (0..3); returns average milliseconds per launch in *ms_out.
VP_API int vp_dbg_gemm_bench(int32_t device, int32_t dtype, int32_t epi, int32_t variant, int32_t group_m, int32_t M,
                             int32_t N, int32_t K, int32_t iters, float* ms_out) {
    if (epi < 0 || epi > 3 || M <= 0 || N <= 0 || K <= 0 || K % 64 || iters <= 0 || !ms_out)
        return fail(nullptr, VP_ERR_INVALID, "bad gemm bench shape");
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    uint16_t *dA, *dW, *dO16 = nullptr;
    float *dB, *dAux = nullptr, *dO32 = nullptr;
    int rc;
    const size_t MN = (size_t)M * N, wrows = pad128(N);
    if ((rc = dalloc(c, &dA, (size_t)M * K)) || (rc = dalloc(c, &dW, wrows * K)) || (rc = dalloc(c, &dB, wrows)) ||
        (rc = dalloc(c, &c->zero, (size_t)256)))
        return dbg_finish(c, rc);
    if (epi >= 2) { if ((rc = dalloc(c, &dO32, MN)) || (rc = dalloc(c, &dAux, (size_t)192 * N))) return dbg_finish(c, rc); }
    else if ((rc = dalloc(c, &dO16, MN))) return dbg_finish(c, rc);
    vp::fill_random16(c->dtype, dA, (size_t)M * K, 1u, nullptr);
    vp::fill_random16(c->dtype, dW, wrows * K, 2u, nullptr);
    hipMemset(dB, 0, wrows * 4);
    if (dO32) hipMemset(dO32, 0, MN * 4);
    if (dAux) hipMemset(dAux, 0, (size_t)192 * N * 4);
    c->gemm_variant[0] = variant & 0xff;
    c->gemm_group_m[0] = group_m;
    c->gemm_ablate = variant >> 8;   // tools only: ablation flags in the high bits
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    void* outp = epi >= 2 ? (void*)dO32 : (void*)dO16;
    const float* aux = epi == 2 ? dO32 : dAux;
    for (int i = 0; i < 2 && !rc; ++i) rc = gemm(c, 0, epi, dA, dW, dB, outp, aux, M, N, K, N);
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters && !rc; ++i) rc = gemm(c, 0, epi, dA, dW, dB, outp, aux, M, N, K, N);
    hipEventRecord(e1, nullptr);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(c, VP_ERR_HIP, "gemm bench kernel failed");
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return dbg_finish(c, rc);
}
#endif  // VP_TOOLS


// ---- production-configuration GEMM taps (tests/test_gpu_gemm_cfgs.py, tools/gemm8_check.py) ----
}  // extern "C"
namespace {
// a GEMM launch on RANDOM device operands in any production configuration: epi = kernels.h GemmEpi 0, 1 (optionally with the
// LayerNorm-consumer fold), 2, 3, 6; flags: 1 persist, 2 out_blocked, 4 a_blocked, 8 reverse, 16 LayerNorm-consumer fold
struct RandCase {
    vp::GemmArgs g{};
    size_t out_bytes = 0, stats_floats = 0;
    void* out[2] = {nullptr, nullptr};
    float* stats[2] = {nullptr, nullptr};
};
int make_rand_case(vp_ctx* c, RandCase& rc, int epi, int flags, int M, int N, int K, int nout) {
    uint16_t *dA, *dW, *dAux16 = nullptr;
    float *dB, *dAux32 = nullptr, *dRow = nullptr, *dS = nullptr;
    int r;
    const size_t MN = (size_t)M * N, wrows = pad128(N);
    if ((r = dalloc(c, &dA, (size_t)M * K)) || (r = dalloc(c, &dW, wrows * K)) || (r = dalloc(c, &c->zero, (size_t)256))) return r;
    vp::fill_random16(c->dtype, dA, (size_t)M * K, 1u, nullptr);
    vp::fill_random16(c->dtype, dW, wrows * K, 2u, nullptr);
    // flags 64 / 128 (tools/clock_power_probe.py, VP_PROBE_SET=operand_bits): the SAME instruction stream on all-zero operands / on operands that are all the
    // constant 0x3c00 (1.0 in fp16) -- how much of a launch's time is the board power limit (the chip clocks by the energy its operand bits toggle)
    if (flags & 64) { HIPCHK(c, hipMemset(dA, 0, (size_t)M * K * 2)); HIPCHK(c, hipMemset(dW, 0, wrows * K * 2)); }
    if (flags & 128) { HIPCHK(c, hipMemsetD16(dA, 0x3c00, (size_t)M * K)); HIPCHK(c, hipMemsetD16(dW, 0x3c00, wrows * K)); }
    std::vector<float> hb(wrows), hs(wrows), hr((size_t)M * 2);
    uint32_t lcg = 12345u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float)((lcg >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hb) v = rnd();
    for (auto& v : hs) v = 4.f * rnd();
    for (size_t i = 0; i < (size_t)M; ++i) { hr[2 * i] = 0.2f * rnd(); hr[2 * i + 1] = 1.f + 0.4f * rnd(); }
    if ((r = upload_f32(c, &dB, hb.data(), wrows))) return r;
    vp::GemmArgs& g = rc.g;
    g.A = dA; g.W = dW; g.bias = dB; g.M = M; g.N = N; g.K = K; g.ldo = N; g.zero = c->zero; g.Kp = c->Kp;
    g.w_rows = (int)wrows;
    g.persist = (flags & 1) != 0; g.out_blocked = (flags & 2) != 0; g.a_blocked = (flags & 4) != 0; g.reverse = (flags & 8) != 0;
    if (flags & 16) {
        if ((r = upload_f32(c, &dRow, hr.data(), (size_t)M * 2)) || (r = upload_f32(c, &dS, hs.data(), wrows))) return r;
        g.rowstat = dRow; g.ln_s = dS;
    }
    if (epi == vp::EPI_BIAS || epi == vp::EPI_BIAS_GELU) {
        rc.out_bytes = MN * 2;
    } else if (epi == vp::EPI_BIAS_RESID || epi == vp::EPI_POS) {
        rc.out_bytes = MN * 4;
        const size_t na = epi == vp::EPI_BIAS_RESID ? MN : (size_t)192 * N;
        if ((r = dalloc(c, &dAux32, na))) return r;
        std::vector<float> ha(na);
        for (auto& v : ha) v = 2.f * rnd();
        HIPCHK(c, hipMemcpy(dAux32, ha.data(), na * 4, hipMemcpyHostToDevice));
        g.aux = dAux32;
    } else if (epi == vp::EPI_BIAS_RESID_LN) {
        rc.out_bytes = MN * 4;   // hi plane + lo plane
        rc.stats_floats = (size_t)M * (N / 64) * 2;
        if ((r = dalloc(c, &dAux16, 2 * MN))) return r;
        vp::fill_random16(c->dtype, dAux16, MN, 3u, nullptr);
        vp::fill_random16(c->dtype, dAux16 + MN, MN, 4u, nullptr);
        g.aux = (const float*)dAux16;
        g.plane = MN;
    } else {
        return fail(c, VP_ERR_INVALID, "unsupported epilogue for the random GEMM case");
    }
    for (int i = 0; i < nout; ++i) {
        char* o;
        if ((r = dalloc(c, &o, rc.out_bytes))) return r;
        HIPCHK(c, hipMemset(o, 0xff, rc.out_bytes));
        rc.out[i] = o;
        if (rc.stats_floats) {
            if ((r = dalloc(c, &rc.stats[i], rc.stats_floats))) return r;
            HIPCHK(c, hipMemset(rc.stats[i], 0xff, rc.stats_floats * 4));
        }
    }
    HIPCHK(c, hipDeviceSynchronize());
    return VP_OK;
}
}  // namespace
extern "C" {

// ONE launch of any production GEMM configuration on HOST fp32 data (tests/test_gpu_gemm_cfgs.py): operands are rounded to
// `dtype` exactly as the packer / producing kernels round them, layouts (64x64-blocked A / output, two-plane residual stream,
// hi+lo final-conv weights) are built and undone here.
//   epi 0 / 1: out[M,N] 16-bit (returned as fp32); rowstat [M,2] + ln_s [N] non-NULL = LayerNorm-consumer fold
//   epi 2 / 3: out[M,N] fp32, aux = residual [M,N] / pos [192,N]
//   epi 6 / 7: aux = fp32 residual [M,N] (split into hi + lo planes on upload) / pos [192,N]; out = hi + lo planes summed;
//              stats [M, N/64, 2] = (sum, centred M2) per 64-column granule
//   epi 5:     W = final 1x1 conv weight [N = Kp, K = 256], A = [M = B 3072, 256]; out = heatmaps [B, Kp, 3072] fp32
// flags: 1 persistent, 2 out_blocked, 4 a_blocked, 8 reverse
VP_API int vp_dbg_gemm_case(int32_t device, int32_t dtype, int32_t epi, int32_t variant, int32_t group_m, int32_t flags, int32_t M,
                            int32_t N, int32_t K, const float* A, const float* W, const float* bias, const float* aux, const float* rowstat,
                            const float* ln_s, float* out, float* stats) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 64 || !A || !W || !bias || !out) return fail(nullptr, VP_ERR_INVALID, "bad gemm case");
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    c->Kp = N;
    int r;
    const size_t MN = (size_t)M * N, wrows = pad128(N);
    const bool ablk = (flags & 4) != 0, oblk = (flags & 2) != 0;
    uint16_t *dA, *dW;
    float *dB, *dAux32 = nullptr, *dRow = nullptr, *dS = nullptr, *dStats = nullptr;
    uint16_t* dAux16 = nullptr;
    void* dOut = nullptr;
    // A (optionally in the 64x64-blocked layout [M/64][K/64][64][64])
    {
        std::vector<uint16_t> ha((size_t)M * K);
        for (size_t m = 0; m < (size_t)M; ++m)
            for (size_t k = 0; k < (size_t)K; ++k) {
                const size_t dst = ablk ? ((((m >> 6) * (K >> 6) + (k >> 6)) << 12) + ((m & 63) << 6) + (k & 63)) : m * K + k;
                ha[dst] = host_to_bits(A[m * K + k], c->dtype);
            }
        if ((r = dalloc(c, &dA, (size_t)M * K))) return dbg_finish(c, r);
        if (hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, "H2D"));
    }
    size_t fin_rows = 0;
    if (epi == vp::EPI_HEATMAP) { if ((r = upload_final(c, &dW, W, N, K, &fin_rows))) return dbg_finish(c, r); }
    else if ((r = upload_mat(c, &dW, W, N, K, wrows))) return dbg_finish(c, r);
    if ((r = upload_f32(c, &dB, bias, N, wrows)) || (r = dalloc(c, &c->zero, (size_t)256))) return dbg_finish(c, r);
    vp::GemmArgs g{};
    g.A = dA; g.W = dW; g.bias = dB; g.M = M; g.N = N; g.K = K; g.ldo = N; g.zero = c->zero; g.Kp = N;
    g.w_rows = (int)wrows; g.variant = variant; g.group_m = group_m;
    g.persist = (flags & 1) != 0; g.out_blocked = oblk; g.a_blocked = ablk; g.reverse = (flags & 8) != 0;
    if (rowstat && ln_s) {
        if ((r = upload_f32(c, &dRow, rowstat, (size_t)M * 2)) || (r = upload_f32(c, &dS, ln_s, N, wrows))) return dbg_finish(c, r);
        g.rowstat = dRow; g.ln_s = dS;
    }
    size_t out_bytes = 0;
    const bool prod = epi == vp::EPI_BIAS_RESID_LN || epi == vp::EPI_POS_LN;
    if (epi == vp::EPI_BIAS || epi == vp::EPI_BIAS_GELU) out_bytes = MN * 2;
    else if (epi == vp::EPI_BIAS_RESID || epi == vp::EPI_POS || prod) out_bytes = MN * 4;
    else if (epi == vp::EPI_HEATMAP) { out_bytes = MN * 4; g.N = (int)fin_rows; g.ldo = 0; g.w_rows = (int)pad128(fin_rows); }
    else return dbg_finish(c, fail(c, VP_ERR_INVALID, "unsupported epilogue"));
    if (epi == vp::EPI_BIAS_RESID || epi == vp::EPI_POS || epi == vp::EPI_POS_LN) {
        if (!aux) return dbg_finish(c, fail(c, VP_ERR_INVALID, "aux required"));
        if ((r = upload_f32(c, &dAux32, aux, epi == vp::EPI_BIAS_RESID ? MN : (size_t)192 * N))) return dbg_finish(c, r);
        g.aux = dAux32;
    }
    if (epi == vp::EPI_BIAS_RESID_LN) {
        if (!aux) return dbg_finish(c, fail(c, VP_ERR_INVALID, "aux required"));
        std::vector<uint16_t> hp(2 * MN);
        for (size_t i = 0; i < MN; ++i) {
            const uint16_t hi = host_to_bits(aux[i], c->dtype);
            hp[i] = hi;
            hp[MN + i] = host_to_bits(aux[i] - host_from_bits(hi, c->dtype), c->dtype);
        }
        if ((r = dalloc(c, &dAux16, 2 * MN))) return dbg_finish(c, r);
        if (hipMemcpy(dAux16, hp.data(), hp.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, "H2D"));
        g.aux = (const float*)dAux16;
    }
    if (prod) {
        g.plane = MN;
        if ((r = dalloc(c, &dStats, (size_t)M * (N / 64) * 2))) return dbg_finish(c, r);
        g.stats_out = dStats;
    }
    char* o;
    if ((r = dalloc(c, &o, out_bytes))) return dbg_finish(c, r);
    dOut = o;
    hipMemset(dOut, 0xff, out_bytes);
    g.out = dOut;
    const int splitk = (flags >> 8) & 15;   // epi 6 only: S partial products (EPI_PARTIAL) + splitk_reduce_kernel instead of the one-launch residual epilogue
    hipError_t e;
    if (splitk > 1) {
        if (epi != vp::EPI_BIAS_RESID_LN) return dbg_finish(c, fail(c, VP_ERR_INVALID, "split-K is a residual-GEMM path (epi 6)"));
        float* ws = nullptr;
        if ((r = dalloc(c, &ws, (size_t)splitk * MN))) return dbg_finish(c, r);
        hipMemset(ws, 0xff, (size_t)splitk * MN * 4);
        vp::GemmArgs p = g;
        p.out = ws; p.aux = nullptr; p.bias = nullptr; p.stats_out = nullptr; p.plane = 0; p.splitk = splitk; p.persist = 0;
        e = vp::gemm_launch(c->dtype, vp::EPI_PARTIAL, p, nullptr);
        if (e == hipSuccess) e = vp::splitk_reduce_launch(c->dtype, ws, splitk, dB, dAux16, MN, dStats, M, N, nullptr);
        dOut = dAux16;   // the reduction updates the residual planes in place
    } else {
        e = vp::gemm_launch(c->dtype, epi, g, nullptr);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("gemm case: ") + hipGetErrorString(e)));
    if (epi == vp::EPI_BIAS || epi == vp::EPI_BIAS_GELU) {
        std::vector<float> t(MN);
        if ((r = download16(c, (const uint16_t*)dOut, t.data(), MN))) return dbg_finish(c, r);
        for (size_t m = 0; m < (size_t)M; ++m)
            for (size_t n = 0; n < (size_t)N; ++n) {
                const size_t src = oblk ? ((((m >> 6) * ((size_t)N >> 6) + (n >> 6)) << 12) + ((m & 63) << 6) + (n & 63)) : m * N + n;
                out[m * N + n] = t[src];
            }
    } else if (prod) {
        std::vector<float> hi(MN), lo(MN);
        if ((r = download16(c, (const uint16_t*)dOut, hi.data(), MN)) || (r = download16(c, (const uint16_t*)dOut + MN, lo.data(), MN))) return dbg_finish(c, r);
        for (size_t i = 0; i < MN; ++i) out[i] = hi[i] + lo[i];
        if (stats && hipMemcpy(stats, dStats, (size_t)M * (N / 64) * 8, hipMemcpyDeviceToHost) != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, "D2H"));
    } else {
        if (hipMemcpy(out, dOut, out_bytes, hipMemcpyDeviceToHost) != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, "D2H"));
    }
    return dbg_finish(c, VP_OK);
}

#ifdef VP_TOOLS   // timing tap of the measurement build (include/vitpose_hip_tools.h)
// average milliseconds per launch of one production GEMM configuration on random operands
VP_API int vp_dbg_gemm_bench2(int32_t device, int32_t dtype, int32_t epi, int32_t variant, int32_t group_m, int32_t flags, int32_t M,
                              int32_t N, int32_t K, int32_t iters, float* ms_out) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 64 || iters <= 0 || !ms_out) return fail(nullptr, VP_ERR_INVALID, "bad gemm bench shape");
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    RandCase rc;
    int r = make_rand_case(c, rc, epi, flags, M, N, K, 1);
    if (r) return dbg_finish(c, r);
    vp::GemmArgs g = rc.g;
    g.variant = variant & 0xff; g.group_m = group_m; g.ablate = variant >> 8;
    g.out = rc.out[0]; g.stats_out = rc.stats[0];
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = vp::gemm_launch(c->dtype, epi, g, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters && e == hipSuccess; ++i) e = vp::gemm_launch(c->dtype, epi, g, nullptr);
    hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("gemm bench2: ") + hipGetErrorString(e)));
    return dbg_finish(c, VP_OK);
}
#endif  // VP_TOOLS

#ifdef VP_TOOLS
// tools/gemm8_timeline.py: one gemm8 launch (variant 16 / 17, epi 0 / 1) with cycle stamps of waves 0 and 4 of every workgroup:
// stamps[wg][group][tile < 16][8] = (main loop begin, main loop end, epilogue end, P4 wait of K-tile 0 begin / end, of K-tile 1 begin / end, 0)
VP_API int vp_dbg_gemm8_timeline(int32_t device, int32_t dtype, int32_t epi, int32_t variant, int32_t flags, int32_t ablate, int32_t M,
                                 int32_t N, int32_t K, uint64_t* stamps, int32_t max_wg) {
    if ((epi != 0 && epi != 1 && epi != vp::EPI_BIAS_RESID_LN) || !stamps || max_wg < 256) return fail(nullptr, VP_ERR_INVALID, "bad timeline request");
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    RandCase rc;
    int r = make_rand_case(c, rc, epi, flags, M, N, K, 1);
    if (r) return dbg_finish(c, r);
    unsigned long long* dS;
    const size_t nst = (size_t)max_wg * 2 * 16 * 8;
    if ((r = dalloc(c, &dS, nst))) return dbg_finish(c, r);
    hipMemset(dS, 0, nst * 8);
    vp::GemmArgs g = rc.g;
    g.variant = variant; g.group_m = 8; g.out = rc.out[0];
    if (epi == vp::EPI_BIAS_RESID_LN) g.stats_out = rc.stats[0];
    hipError_t e = vp::gemm_launch(c->dtype, epi, g, nullptr);   // warm
    g.ablate = 32 | ablate;
    if (epi == vp::EPI_BIAS_RESID_LN) { g.stats_out = rc.stats[0]; g.ln_part = (const float*)dS; }   // the residual GEMM writes real statistics: stamps go to the unused ln_part
    else g.stats_out = (float*)dS;
    if (e == hipSuccess) e = vp::gemm_launch(c->dtype, epi, g, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(stamps, dS, nst * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("gemm8 timeline: ") + hipGetErrorString(e)));
    return dbg_finish(c, VP_OK);
}
#endif  // VP_TOOLS

#ifdef VP_TOOLS   // timing tap of the measurement build (include/vitpose_hip_tools.h)
// run two configurations of the same GEMM on the same random operands `reps` times each and compare every output byte
// (and the row statistics): the race / schedule screen for kernels whose arithmetic order is identical by construction
VP_API int vp_dbg_gemm_compare(int32_t device, int32_t dtype, int32_t epi, int32_t variant_a, int32_t group_a, int32_t flags_a,
                               int32_t variant_b, int32_t group_b, int32_t flags_b, int32_t M, int32_t N, int32_t K, int32_t reps,
                               uint64_t* n_mismatch, double* max_abs_diff) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 64 || reps <= 0 || !n_mismatch || !max_abs_diff) return fail(nullptr, VP_ERR_INVALID, "bad gemm compare shape");
    if ((flags_a & (2 | 4 | 16)) != (flags_b & (2 | 4 | 16))) return fail(nullptr, VP_ERR_INVALID, "layout / fold flags must agree");
    vp_ctx* c = dbg_ctx(device, dtype);
    if (!c) return VP_ERR_HIP;
    RandCase rc;
    int r = make_rand_case(c, rc, epi, flags_a, M, N, K, 2);
    if (r) return dbg_finish(c, r);
    *n_mismatch = 0; *max_abs_diff = 0.0;
    std::vector<uint16_t> ha(rc.out_bytes / 2), hb2(rc.out_bytes / 2);
    std::vector<float> sa(rc.stats_floats), sb(rc.stats_floats);
    for (int rep = 0; rep < reps; ++rep) {
        for (int w = 0; w < 2; ++w) {
            vp::GemmArgs g = rc.g;
            const int fl = w ? flags_b : flags_a;
            g.variant = w ? variant_b : variant_a; g.group_m = w ? group_b : group_a;
            g.persist = (fl & 1) != 0; g.reverse = (fl & 8) != 0;
            g.out = rc.out[w]; g.stats_out = rc.stats[w];
            hipMemsetAsync(rc.out[w], 0xff, rc.out_bytes, nullptr);
            hipError_t e = vp::gemm_launch(c->dtype, epi, g, nullptr);
            if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("gemm compare launch ") + (w ? "B: " : "A: ") + hipGetErrorString(e)));
        }
        hipError_t e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(ha.data(), rc.out[0], rc.out_bytes, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(hb2.data(), rc.out[1], rc.out_bytes, hipMemcpyDeviceToHost);
        if (e == hipSuccess && rc.stats_floats) e = hipMemcpy(sa.data(), rc.stats[0], rc.stats_floats * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && rc.stats_floats) e = hipMemcpy(sb.data(), rc.stats[1], rc.stats_floats * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, std::string("gemm compare: ") + hipGetErrorString(e)));
        const bool f32out = (epi == vp::EPI_BIAS_RESID || epi == vp::EPI_POS);
        if (f32out) {
            const float* fa = (const float*)ha.data(); const float* fb = (const float*)hb2.data();
            for (size_t i = 0; i < rc.out_bytes / 4; ++i)
                if (std::memcmp(&fa[i], &fb[i], 4)) { ++*n_mismatch; const double d = std::fabs((double)fa[i] - (double)fb[i]); if (!(d <= *max_abs_diff)) *max_abs_diff = d; }
        } else {
            for (size_t i = 0; i < ha.size(); ++i)
                if (ha[i] != hb2[i]) {
                    ++*n_mismatch;
                    const double d = std::fabs((double)host_from_bits(ha[i], c->dtype) - (double)host_from_bits(hb2[i], c->dtype));
                    if (!(d <= *max_abs_diff)) *max_abs_diff = d;
                }
        }
        for (size_t i = 0; i < sa.size(); ++i)
            if (std::memcmp(&sa[i], &sb[i], 4)) { ++*n_mismatch; const double d = std::fabs((double)sa[i] - (double)sb[i]); if (!(d <= *max_abs_diff)) *max_abs_diff = d; }
    }
    return dbg_finish(c, VP_OK);
}
#endif  // VP_TOOLS

// frame + crop geometry -> the uint8 [n,256,192,3] crops the model is fed (device crop/pad/resize kernel alone)
VP_API int vp_dbg_crop_prep(int32_t device, const uint8_t* frame, int32_t fh, int32_t fw, const int32_t* crop_params, int32_t n, uint8_t* out) {
    if (!frame || !crop_params || !out || n <= 0 || fh <= 0 || fw <= 0) return fail(nullptr, VP_ERR_INVALID, "bad argument");
    vp_ctx* c = dbg_ctx(device, VP_DTYPE_F16);
    if (!c) return VP_ERR_HIP;
    uint8_t *df, *dout;
    int32_t* dp;
    int rc;
    const size_t fb = (size_t)fh * fw * 3, ob = (size_t)n * 256 * 192 * 3;
    if ((rc = dalloc(c, &df, fb)) || (rc = dalloc(c, &dout, ob)) || (rc = dalloc(c, &dp, (size_t)n * 8))) return dbg_finish(c, rc);
    hipError_t e = hipMemcpy(df, frame, fb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dp, crop_params, (size_t)n * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = vp::crop_resize_launch(df, fh, fw, dp, dout, n, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, dout, ob, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(c, VP_ERR_HIP, std::string("crop_prep: ") + hipGetErrorString(e));
    return dbg_finish(c, rc);
}

// BASELINE config 5 probe: rows quantised to OCP e4m3 on device + one GEMM through v_mfma_f32_16x16x128_f8f6f4 (fp8_probe.hip)
VP_API int vp_dbg_fp8_gemm(int32_t device, int32_t M, int32_t N, int32_t K, const float* A, const float* a_scale, const float* W,
                           const float* w_scale, float* out, uint8_t* a_codes, uint8_t* w_codes) {
    if (M <= 0 || N <= 0 || K <= 0 || M % 16 || N % 16 || K % 128 || !A || !W || !a_scale || !w_scale || !out)
        return fail(nullptr, VP_ERR_INVALID, "bad fp8 probe shape");
    vp_ctx* c = dbg_ctx(device, VP_DTYPE_F16);
    if (!c) return VP_ERR_HIP;
    float *dA, *dW, *dAs, *dWs, *dO;
    uint8_t *dA8, *dW8;
    int rc;
    if ((rc = upload_f32(c, &dA, A, (size_t)M * K)) || (rc = upload_f32(c, &dW, W, (size_t)N * K)) || (rc = upload_f32(c, &dAs, a_scale, M)) ||
        (rc = upload_f32(c, &dWs, w_scale, N)) || (rc = dalloc(c, &dO, (size_t)M * N)) || (rc = dalloc(c, &dA8, (size_t)M * K)) ||
        (rc = dalloc(c, &dW8, (size_t)N * K)))
        return dbg_finish(c, rc);
    hipError_t e = vp::fp8_probe_launch(dA, dW, dAs, dWs, dA8, dW8, dO, M, N, K, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, dO, (size_t)M * N * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && a_codes) e = hipMemcpy(a_codes, dA8, (size_t)M * K, hipMemcpyDeviceToHost);
    if (e == hipSuccess && w_codes) e = hipMemcpy(w_codes, dW8, (size_t)N * K, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(c, VP_ERR_HIP, std::string("fp8 probe: ") + hipGetErrorString(e));
    return dbg_finish(c, rc);
}

// MX probe (round 4): A -> MXFP8 on device (mx8.h layouts), W -> e4m3 with the per-row scale given; out = block-scaled MFMA product.
// a_codes [M*K] (blocked layout), a_scales [M*K/32] (packed dword layout), w_codes [N*K] may be NULL.
VP_API int vp_dbg_mx_gemm(int32_t device, int32_t M, int32_t N, int32_t K, const float* A, const float* W, const float* w_scale, float* out,
                          uint8_t* a_codes, uint8_t* a_scales, uint8_t* w_codes) {
    if (M <= 0 || N <= 0 || K <= 0 || M % 64 || N % 16 || K % 128 || !A || !W || !w_scale || !out) return fail(nullptr, VP_ERR_INVALID, "bad mx probe shape");
    vp_ctx* c = dbg_ctx(device, VP_DTYPE_F16);
    if (!c) return VP_ERR_HIP;
    float *dA, *dW, *dWs, *dO;
    uint8_t *dA8, *dAs, *dW8;
    int rc;
    if ((rc = upload_f32(c, &dA, A, (size_t)M * K)) || (rc = upload_f32(c, &dW, W, (size_t)N * K)) || (rc = upload_f32(c, &dWs, w_scale, N)) ||
        (rc = dalloc(c, &dO, (size_t)M * N)) || (rc = dalloc(c, &dA8, (size_t)M * K)) || (rc = dalloc(c, &dAs, (size_t)M * K / 32)) ||
        (rc = dalloc(c, &dW8, (size_t)N * K)))
        return dbg_finish(c, rc);
    hipError_t e = vp::mx_probe_launch(dA, dW, dWs, dA8, dAs, dW8, dO, M, N, K, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, dO, (size_t)M * N * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && a_codes) e = hipMemcpy(a_codes, dA8, (size_t)M * K, hipMemcpyDeviceToHost);
    if (e == hipSuccess && a_scales) e = hipMemcpy(a_scales, dAs, (size_t)M * K / 32, hipMemcpyDeviceToHost);
    if (e == hipSuccess && w_codes) e = hipMemcpy(w_codes, dW8, (size_t)N * K, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(c, VP_ERR_HIP, std::string("mx probe: ") + hipGetErrorString(e));
    return dbg_finish(c, rc);
}

// ONE launch of the MXFP8 GEMM kernel (gemm8f.hip) on host fp32 data (tests/test_gpu_fp8.py).  A [M,K] is quantised to MXFP8 on device
// (mx_quantize_launch: the layouts of csrc/mx8.h), W [N,K] on the host exactly as the weight packer does (per-output-channel scale);
// a_deq / w_deq return what the codes and scales stand for, so that the test can restate the product exactly.
//   epi 0: out = a.w^T * w_scale + bias, rounded to fp16        epi 1: out = gelu(...) as MXFP8 (returned de-quantised)
//   epi 6: out = ... + aux (two-plane residual, returned as hi + lo), stats [M, N/64, 2]
VP_API int vp_dbg_gemm_fp8_case(int32_t device, int32_t epi, int32_t M, int32_t N, int32_t K, const float* A, const float* W, const float* bias,
                                const float* aux, float* out, float* stats, float* a_deq, float* w_deq) {
    if (M <= 0 || N <= 0 || K <= 0 || M % 256 || K % 256 || N % 64 || !A || !W || !bias || !out || (epi != 0 && epi != 1 && epi != 6))
        return fail(nullptr, VP_ERR_INVALID, "bad fp8 gemm case");
    vp_ctx* c = dbg_ctx(device, VP_DTYPE_F16);
    if (!c) return VP_ERR_HIP;
    int r;
    const size_t MN = (size_t)M * N, MK = (size_t)M * K;
    float *dA, *dB, *dWs, *dStats = nullptr;
    uint8_t *dA8, *dAs, *dW8, *dOs = nullptr;
    uint16_t* dAux16 = nullptr;
    char* dOut;
    if ((r = upload_f32(c, &dA, A, MK)) || (r = dalloc(c, &dA8, MK)) || (r = dalloc(c, &dAs, MK / 32)) ||
        (r = upload_fp8_rows(c, &dW8, &dWs, nullptr, W, nullptr, nullptr, nullptr, N, K)) || (r = upload_f32(c, &dB, bias, N, pad128(N))))
        return dbg_finish(c, r);
    hipError_t e = vp::mx_quantize_launch(dA, dA8, dAs, M, K, nullptr);
    if (e != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, "mx quantize"));
    const size_t out_bytes = epi == 0 ? MN * 2 : epi == 1 ? MN : MN * 4;
    if ((r = dalloc(c, &dOut, out_bytes))) return dbg_finish(c, r);
    hipMemset(dOut, 0xff, out_bytes);
    LnFuse ln;
    if (epi == 1 && (r = dalloc(c, &dOs, MN / 32))) return dbg_finish(c, r);
    if (epi == 6) {
        if (!aux) return dbg_finish(c, fail(c, VP_ERR_INVALID, "aux required"));
        std::vector<uint16_t> hp(2 * MN);
        for (size_t i = 0; i < MN; ++i) {
            const uint16_t hi = host_to_bits(aux[i], c->dtype);
            hp[i] = hi;
            hp[MN + i] = host_to_bits(aux[i] - host_from_bits(hi, c->dtype), c->dtype);
        }
        if ((r = dalloc(c, &dAux16, 2 * MN)) || (r = dalloc(c, &dStats, (size_t)M * (N / 64) * 2))) return dbg_finish(c, r);
        if (hipMemcpy(dAux16, hp.data(), hp.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, "H2D"));
        ln.plane = MN; ln.stats_out = dStats;
    }
    r = gemm_fp8(c, epi == 0 ? VP_PROF_GEMM_QKV : epi == 1 ? VP_PROF_GEMM_FC1 : VP_PROF_GEMM_FC2, epi, dA8, dAs, dW8, dWs, dB, dOut, dOs,
                 (const float*)dAux16, M, N, K, &ln);
    if (!r && hipDeviceSynchronize() != hipSuccess) r = fail(c, VP_ERR_HIP, "fp8 gemm kernel failed");
    if (r) return dbg_finish(c, r);
    // what the operands stand for
    {
        std::vector<uint8_t> ca(MK), sa(MK / 32);
        if (hipMemcpy(ca.data(), dA8, MK, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(sa.data(), dAs, MK / 32, hipMemcpyDeviceToHost) != hipSuccess)
            return dbg_finish(c, fail(c, VP_ERR_HIP, "D2H"));
        if (a_deq)
            for (size_t m = 0; m < (size_t)M; ++m)
                for (size_t k = 0; k < (size_t)K; ++k)
                    a_deq[m * K + k] = vp_host_e4m3_to_float(ca[vp::mx_code_off(m, k, K)]) * std::ldexp(1.0f, (int)sa[vp::mx_scale_off(m, k >> 5, K)] - 127);
        if (w_deq) {
            std::vector<uint8_t> cw((size_t)N * K);
            std::vector<float> sw(N);
            if (hipMemcpy(cw.data(), dW8, cw.size(), hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(sw.data(), dWs, (size_t)N * 4, hipMemcpyDeviceToHost) != hipSuccess)
                return dbg_finish(c, fail(c, VP_ERR_HIP, "D2H"));
            for (size_t n = 0; n < (size_t)N; ++n)
                for (size_t k = 0; k < (size_t)K; ++k) w_deq[n * K + k] = vp_host_e4m3_to_float(cw[n * K + k]) * sw[n];
        }
    }
    if (epi == 0) {
        r = download16(c, (const uint16_t*)dOut, out, MN);
    } else if (epi == 1) {
        std::vector<uint8_t> co(MN), so(MN / 32);
        if (hipMemcpy(co.data(), dOut, MN, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(so.data(), dOs, MN / 32, hipMemcpyDeviceToHost) != hipSuccess)
            return dbg_finish(c, fail(c, VP_ERR_HIP, "D2H"));
        for (size_t m = 0; m < (size_t)M; ++m)
            for (size_t n = 0; n < (size_t)N; ++n)
                out[m * N + n] = vp_host_e4m3_to_float(co[vp::mx_code_off(m, n, N)]) * std::ldexp(1.0f, (int)so[vp::mx_scale_off(m, n >> 5, N)] - 127);
    } else {
        std::vector<float> hi(MN), lo(MN);
        if ((r = download16(c, (const uint16_t*)dOut, hi.data(), MN)) || (r = download16(c, (const uint16_t*)dOut + MN, lo.data(), MN))) return dbg_finish(c, r);
        for (size_t i = 0; i < MN; ++i) out[i] = hi[i] + lo[i];
        if (stats && hipMemcpy(stats, dStats, (size_t)M * (N / 64) * 8, hipMemcpyDeviceToHost) != hipSuccess) return dbg_finish(c, fail(c, VP_ERR_HIP, "D2H"));
    }
    return dbg_finish(c, r);
}

// host-only: fp32 -> OCP e4m3 codes with the library's own converter (the one the weight packer of the fp8 mode uses)
VP_API int vp_dbg_host_e4m3(const float* in, uint8_t* out, int64_t n) {
    if (!in || !out || n < 0) return VP_ERR_INVALID;
    for (int64_t i = 0; i < n; ++i) out[i] = vp_host_e4m3(in[i]);
    return VP_OK;
}

#ifdef VP_TOOLS
VP_API int vp_dbg_hwid_probe(int32_t device, int32_t blocks, int32_t threads, int32_t lds_bytes, int32_t spin, uint32_t* out) {
    if (!out || blocks <= 0 || blocks > 65536 || threads <= 0 || threads > 1024 || lds_bytes < 16 || lds_bytes > 160 * 1024 || spin < 0) return fail(nullptr, VP_ERR_INVALID, "bad argument");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, VP_ERR_HIP, "no HIP device");
    uint32_t* d = nullptr;
    if (hipMalloc((void**)&d, (size_t)blocks * 16) != hipSuccess) return fail(nullptr, VP_ERR_HIP, "hipMalloc");
    hipError_t e = vp::hwid_probe_launch(d, blocks, threads, lds_bytes, spin, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    hipFree(d);
    return e == hipSuccess ? VP_OK : fail(nullptr, VP_ERR_HIP, hipGetErrorString(e));
}
#endif

// Calibration: kind 0/1 = MFMA-only loop (16x16x32 / 32x32x16 f16) in TFLOP/s, 2 = float4 copy in TB/s (read+write).
#ifdef VP_TOOLS   // timing tap of the measurement build (include/vitpose_hip_tools.h)
VP_API int vp_dbg_peak(int32_t device, int32_t kind, double* result) {
    const bool known = (kind >= 0 && kind <= 12) || (kind >= 100 && kind < 164) || (kind >= 170 && kind < 178) || (kind >= 200 && kind < 248) ||
                       (kind >= 300 && kind < 492) || (kind >= 500 && kind < 504);
    if (!result || !known) return fail(nullptr, VP_ERR_INVALID, "bad argument");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, VP_ERR_HIP, "no HIP device");
    hipError_t e = vp::peak_bench(kind, result);
    return e == hipSuccess ? VP_OK : fail(nullptr, VP_ERR_HIP, hipGetErrorString(e));
}
#endif  // VP_TOOLS

}  // extern "C"
