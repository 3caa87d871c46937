// mtk_capi.hip — context lifetime and error reporting of libmtk_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "mtk_common.h"

static thread_local char g_err[512] = "";

void mtk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mtk_last_error(void) { return g_err; }
extern "C" int mtk_version(void) { return 100; }

void mtk_free_mlp(mtk_ctx* c, bool keep_ws);  // mtk_mlp_f32.hip
int mtk_noalloc_init(int device);  // mtk_mlp_f32.hip: resolves this device's copy of the static scratch array

extern "C" int mtk_ctx_create(int device, int T, int Nn, int C, const int32_t* K_host, const float* logK_host,
                              const double* tables_host, mtk_ctx** out) {
    if (!out) MTK_FAIL(MTK_ERR_ARG, "mtk_ctx_create: out is null");
    *out = nullptr;
    if (T < 1 || Nn < 0 || C < 0 || !tables_host) MTK_FAIL(MTK_ERR_ARG, "mtk_ctx_create: bad T/Nn/C or null tables");
    if (C > MTK_C_MAX) MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_ctx_create: C = %d exceeds %d", C, MTK_C_MAX);
    if (C > 0 && (!K_host || !logK_host)) MTK_FAIL(MTK_ERR_ARG, "mtk_ctx_create: K_host/logK_host required when C > 0");
    if (C > 0 && K_host[0] == 0) C = 0;  // the reference's "num_classes = [0]" spelling of "no categorical block"
    int n_dev = 0;
    MTK_HIP(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) MTK_FAIL(MTK_ERR_ARG, "mtk_ctx_create: device %d of %d", device, n_dev);
    MTK_HIP(hipSetDevice(device));
    {
        const int rc = mtk_noalloc_init(device);  // once per device, with that device current
        if (rc) return rc;
    }

    mtk_ctx* c = new mtk_ctx();
    c->device = device;
    c->T = T;
    c->Nn = Nn;
    c->C = C;
    c->off.assign(1, 0);
    for (int k = 0; k < C; ++k) {
        if (K_host[k] < 1) {
            delete c;
            MTK_FAIL(MTK_ERR_ARG, "mtk_ctx_create: K[%d] = %d", k, K_host[k]);
        }
        c->K.push_back(K_host[k]);
        c->logK.push_back(logK_host[k]);
        c->off.push_back(c->off.back() + K_host[k]);
    }
    c->S = c->off.back();
    c->D = Nn + c->S;
    c->tables.assign(tables_host, tables_host + (size_t)MTK_NTAB * T);

    std::vector<SchedRow> rows(T);
    for (int t = 0; t < T; ++t) {
        rows[t].sqrt_acp = (float)c->tab(MTK_TAB_SQRT_ACP, t);
        rows[t].sqrt_1m_acp = (float)c->tab(MTK_TAB_SQRT_1M_ACP, t);
        rows[t].log_alpha = (float)c->tab(MTK_TAB_LOG_ALPHA, t);
        rows[t].log_1m_alpha = (float)c->tab(MTK_TAB_LOG_1M_ALPHA, t);
        rows[t].log_acp = (float)c->tab(MTK_TAB_LOG_ACP, t);
        rows[t].log_1m_acp = (float)c->tab(MTK_TAB_LOG_1M_ACP, t);
        rows[t].sqrt_recip_acp = (float)c->tab(MTK_TAB_SQRT_RECIP_ACP, t);
        rows[t].sqrt_recipm1_acp = (float)c->tab(MTK_TAB_SQRT_RECIPM1_ACP, t);
    }
    std::vector<float> lk(c->logK);
    if (lk.empty()) lk.push_back(0.0f);
    hipError_t e = hipMalloc(&c->d_rows, sizeof(SchedRow) * T);
    if (e == hipSuccess) e = hipMalloc(&c->d_off, sizeof(int32_t) * c->off.size());
    if (e == hipSuccess) e = hipMalloc(&c->d_logK, sizeof(float) * lk.size());
    if (e == hipSuccess) e = hipMemcpy(c->d_rows, rows.data(), sizeof(SchedRow) * T, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->d_off, c->off.data(), sizeof(int32_t) * c->off.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->d_logK, lk.data(), sizeof(float) * lk.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        mtk_ctx_destroy(c);
        MTK_FAIL(MTK_ERR_HIP, "mtk_ctx_create: device allocation failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return MTK_OK;
}

extern "C" int64_t mtk_ctx_workspace_generation(mtk_ctx* c) { return c ? (int64_t)c->ws_gen : -1; }

extern "C" int64_t mtk_ctx_alloc_count(mtk_ctx* c) { return c ? (int64_t)c->n_malloc : -1; }

int mtk_alloc(mtk_ctx* c, void** p, size_t bytes) {
    for (size_t i = 0; i < c->pool.size(); ++i) {
        if (c->pool[i].bytes != bytes) continue;
        *p = c->pool[i].p;
        c->pool.erase(c->pool.begin() + (long)i);
        c->alloc_bytes[*p] = bytes;
        return MTK_OK;
    }
    *p = nullptr;
    MTK_HIP(hipMalloc(p, bytes));
    ++c->n_malloc;
    c->alloc_bytes[*p] = bytes;
    return MTK_OK;
}

void mtk_release(mtk_ctx* c, void* p) {
    if (!p) return;
    auto it = c->alloc_bytes.find(p);
    if (it != c->alloc_bytes.end() && c->pooling) {
        c->pool.push_back({p, it->second, c->pool_gen});
        c->alloc_bytes.erase(it);
        return;
    }
    if (it != c->alloc_bytes.end()) c->alloc_bytes.erase(it);
    (void)hipFree(p);
}

bool mtk_pool_has_stale(const mtk_ctx* c, uint64_t min_gen) {
    for (auto& b : c->pool) if (b.gen < min_gen) return true;
    return false;
}

void mtk_pool_drain(mtk_ctx* c, uint64_t min_gen) {
    size_t keep = 0;
    for (auto& b : c->pool) {
        if (b.gen < min_gen) (void)hipFree(b.p);
        else c->pool[keep++] = b;
    }
    c->pool.resize(keep);
}

extern "C" void mtk_ctx_destroy(mtk_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    c->pooling = false;
    mtk_free_mlp(c, false);
    mtk_pool_drain(c, UINT64_MAX);
    if (c->d_rows) (void)hipFree(c->d_rows);
    if (c->d_off) (void)hipFree(c->d_off);
    if (c->d_logK) (void)hipFree(c->d_logK);
    if (c->d_kl) (void)hipFree(c->d_kl);
    if (c->d_label) (void)hipFree(c->d_label);
    delete c;
}
